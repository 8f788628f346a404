#!/bin/bash
# round-4 GPU call V: the full bench line (all legs) of the final tree (call U's copy was overwritten by the profiled run's line)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4v_bench_full.json 2> gpurun_out/r4v_bench_full.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4v_bench_full.json") if l.startswith("{")][-1])
print("ms/step", round(d["ms_per_step"], 2), "value", round(d["value"], 2), "dense", round(d["roofline"]["frac"], 3), "traffic", d["roofline"]["traffic"])
print({k: (v["launches"], round(v["total_ms"], 2), round(v["tflops"])) for k, v in d["roofline"]["per_kernel"].items()})
print("xblock", {k: (round(v["ms"], 4), round(v["frac"], 3), round(v.get("three_launch_ms", 0), 4)) for k, v in d["cross_attention_block"].items() if isinstance(v, dict)})
for k in ("graph_replay", "batched_2_requests", "plain_pass", "end_to_end", "cpu_baseline", "parity"):
    print("  ", k, json.dumps(d.get(k))[:900])
PY
