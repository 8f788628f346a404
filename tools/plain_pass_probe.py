import json, os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import bench, end_to_end as e2e
from rich_text_to_image_amd.engine import Engine, SDXL_CONFIG
eng = Engine(SDXL_CONFIG, 128, 128, device=0, max_streams=8, max_prompts=8)
eng.init_random_weights(0)
inp = bench.synth_inputs(1000, 4, 128, "cuda:0")
r = e2e.plain_pass(eng, inp, 128)
print(json.dumps({k: r[k] for k in r if k not in ("note",)}))
