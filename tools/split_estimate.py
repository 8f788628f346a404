import json, sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from rich_text_to_image_amd.engine import Engine, SDXL_CONFIG
eng = Engine(SDXL_CONFIG, 128, 128, device=0, max_streams=8, max_prompts=8)
eng.init_random_weights(0)
inp = bench.synth_inputs(1000, 4, 128, "cuda:0")
ts, sig, init_sigma = bench.euler_tables(50)
eng.set_prompts(inp["emb"], inp["pooled"], inp["tid"]); eng.set_masks(inp["masks"]); eng.set_fontsize(torch.tensor([5, 6]), torch.tensor([20.0, 20.0]))
eng.set_schedule(0, ts, sig, 50); eng.set_latents((inp["lat"] * init_sigma).to("cuda:0"))
def t(fn, n=5):
    fn(); fn(); eng.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    eng.synchronize(); torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for label, idx in (("injected", 0), ("plain", 49)):
    full = t(lambda: (eng.region_step_part(idx, 5.0, 0.5, 0.0, True, 0, 1)))
    p0 = t(lambda: eng.region_step_part(idx, 5.0, 0.5, 0.0, True, 0, 2)); p1 = t(lambda: eng.region_step_part(idx, 5.0, 0.5, 0.0, True, 1, 2))
    print(label, "all 7 streams %.2f ms | part 0 %s %.2f ms | part 1 %s %.2f ms" % (full, eng.region_step_part(idx, 5.0, 0.5, 0.0, True, 0, 2)[:2], p0, eng.region_step_part(idx, 5.0, 0.5, 0.0, True, 1, 2)[:2], p1))
