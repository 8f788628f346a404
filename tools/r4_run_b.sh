#!/bin/bash
# round-4 GPU call B: one-pass attention-store kernel (parity + plain-pass timing), stamp breakdown of the fused cross-attention kernel
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_attn_store_gpu.py tests/test_kernels_gpu.py -x -q -k "store or probs_avg or processor or cross" > gpurun_out/r4b_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4b_tests.log
tail -3 gpurun_out/r4b_tests.log
timeout 300 tools/probes/xattn_bench > gpurun_out/r4b_xattn_probe.txt 2>&1; cat gpurun_out/r4b_xattn_probe.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4b_bench.json 2> gpurun_out/r4b_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
for f in ("r4b_bench",):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        print(f, "ms/step", round(d["ms_per_step"], 2), "dense", round(d["roofline"]["frac"], 3))
        for k in ("graph_replay", "batched_2_requests", "plain_pass", "end_to_end"):
            if k in d: print("  ", k, json.dumps(d[k])[:900])
    except Exception as ex:
        print(f, "ERR", ex)
PY
