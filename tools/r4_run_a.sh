#!/bin/bash
# round-4 GPU call A: fused cross-attention parity, headline bench with all legs, same-box A/B against the three-launch form
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "cross_attn or cross_attention" > gpurun_out/r4a_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4a_tests.log
tail -3 gpurun_out/r4a_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err; echo "bench rc=$?"
RTDIFF_DEBUG_FLAGS=16 timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r4a_bench_nofuse.json 2> gpurun_out/r4a_bench_nofuse.err; echo "nofuse rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r4a_bench_fuse2.json 2> gpurun_out/r4a_bench_fuse2.err; echo "fuse2 rc=$?"
python - <<'PY'
import json
for f in ("r4a_bench", "r4a_bench_nofuse", "r4a_bench_fuse2"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        print(f, "ms/step", round(d["ms_per_step"], 2), "dense", round(d["roofline"]["frac"], 3), "xblock", {k: (round(v["ms"], 4), round(v["frac"], 3), round(v.get("three_launch_ms", 0), 4)) for k, v in (d.get("cross_attention_block") or {}).items() if isinstance(v, dict)})
        for k in ("graph_replay", "batched_2_requests", "plain_pass", "end_to_end"):
            if k in d: print("  ", k, json.dumps(d[k])[:600])
    except Exception as ex:
        print(f, "ERR", ex)
PY
