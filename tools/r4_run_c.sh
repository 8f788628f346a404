#!/bin/bash
# round-4 GPU call C: conv K order A/B, fused cross-attention v2 (asm O writes) A/B, attn_store16 with raw barriers
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_attn_store_gpu.py tests/test_kernels_gpu.py -x -q -k "store or probs_avg or cross or conv3x3_on" > gpurun_out/r4c_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4c_tests.log
tail -3 gpurun_out/r4c_tests.log
timeout 300 tools/probes/xattn_bench > gpurun_out/r4c_xattn_probe.txt 2>&1; cat gpurun_out/r4c_xattn_probe.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4c_bench.json 2> gpurun_out/r4c_bench.err; echo "bench rc=$?"
RTDIFF_LIB_PATH=$PWD/rich-text-to-image_amd/librtdiff_tapmajor.so timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r4c_bench_tapmajor.json 2> gpurun_out/r4c_bench_tapmajor.err; echo "tapmajor rc=$?"
RTDIFF_DEBUG_FLAGS=16 timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r4c_bench_nofuse.json 2> gpurun_out/r4c_bench_nofuse.err; echo "nofuse rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r4c_bench_2.json 2> gpurun_out/r4c_bench_2.err; echo "again rc=$?"
python - <<'PY'
import json
for f in ("r4c_bench", "r4c_bench_tapmajor", "r4c_bench_nofuse", "r4c_bench_2"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        pk = d["roofline"]["per_kernel"]
        print(f, "ms/step", round(d["ms_per_step"], 2), "dense", round(d["roofline"]["frac"], 3), {k: (round(v["total_ms"], 2), round(v["tflops"])) for k, v in pk.items()},
              "xblock", {k: (round(v["ms"], 4), round(v["frac"], 3), round(v.get("three_launch_ms", 0), 4)) for k, v in (d.get("cross_attention_block") or {}).items() if isinstance(v, dict)})
        for k in ("plain_pass", "end_to_end"):
            if k in d: print("  ", k, json.dumps(d[k])[:700])
    except Exception as ex:
        print(f, "ERR", ex)
PY
