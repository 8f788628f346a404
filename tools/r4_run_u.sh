#!/bin/bash
# round-4 final measurement call (tree with grouped launches, s_setprio, small-batch tiles): bench line (all legs), rocprof kernel stats of the same command, PMC passes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_attn_store_gpu.py tests/test_sample_gpu.py -x -q > gpurun_out/r4u_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r4u_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4u_bench_full.json 2> gpurun_out/r4u_bench_full.err; echo "bench rc=$?"
timeout 900 bash tools/profile_step.sh r4u --steps 10 --warmup 3 --no-extras > gpurun_out/r4u_profile.log 2>&1; echo "profile rc=$?"
timeout 1500 bash tools/pmc_passes.sh r4u > gpurun_out/r4u_pmc.log 2>&1; echo "pmc rc=$?"; tail -4 gpurun_out/r4u_pmc.log
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4u_bench_full.json") if l.startswith("{")][-1])
print("ms/step", round(d["ms_per_step"], 2), "value", round(d["value"], 2), "dense", round(d["roofline"]["frac"], 3), "traffic", d["roofline"]["traffic"])
print({k: (v["launches"], round(v["total_ms"], 2), round(v["tflops"])) for k, v in d["roofline"]["per_kernel"].items()})
print("xblock", {k: (round(v["ms"], 4), round(v["frac"], 3), round(v.get("three_launch_ms", 0), 4)) for k, v in d["cross_attention_block"].items() if isinstance(v, dict)})
for k in ("graph_replay", "batched_2_requests", "plain_pass", "end_to_end", "cpu_baseline", "parity"):
    print("  ", k, json.dumps(d.get(k))[:700])
PY
