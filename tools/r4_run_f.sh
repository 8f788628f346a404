#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_attn_store_gpu.py tests/test_kernels_gpu.py -x -q -k "store or probs_avg or processor" > gpurun_out/r4f_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4f_tests.log
tail -3 gpurun_out/r4f_tests.log
timeout 300 python tools/attn_store_bench.py > gpurun_out/r4f_attn_store_bench.txt 2>&1; cat gpurun_out/r4f_attn_store_bench.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4f_bench.json 2> gpurun_out/r4f_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4f_bench.json") if l.startswith("{")][-1])
print("ms/step", round(d["ms_per_step"], 2), "dense", round(d["roofline"]["frac"], 3))
for k in ("plain_pass", "end_to_end"):
    print("  ", k, json.dumps(d[k])[:1100])
PY
