#!/bin/bash
# round-4 GPU call R: small-batch tiles (variants 9 - 12): class equality tests, plain pass and SD-v1.5 configs against the round-4
# HEAD library on the same box, config-3 sanity
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
HEADLIB=$PWD/rich-text-to-image_amd/librtdiff_head.so
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm16 or grouped or fp16_trunk" > gpurun_out/r4r_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4r_tests.log
timeout 300 python tools/plain_profile.py --steps 20 > gpurun_out/r4r_plain_new.txt 2>&1; tail -1 gpurun_out/r4r_plain_new.txt
RTDIFF_ALLOW_MISSING_SYMBOLS=1 RTDIFF_LIB_PATH=$HEADLIB timeout 300 python tools/plain_profile.py --steps 20 > gpurun_out/r4r_plain_head.txt 2>&1; tail -1 gpurun_out/r4r_plain_head.txt
timeout 300 python tools/plain_profile.py --steps 20 > gpurun_out/r4r_plain_new2.txt 2>&1; tail -1 gpurun_out/r4r_plain_new2.txt
for c in 1 2; do
  timeout 600 python bench.py --config $c --no-cpu-baseline > gpurun_out/r4r_cfg${c}_new.json 2> gpurun_out/r4r_cfg${c}_new.err; echo "cfg$c new rc=$?"
  RTDIFF_ALLOW_MISSING_SYMBOLS=1 RTDIFF_LIB_PATH=$HEADLIB timeout 600 python bench.py --config $c --no-cpu-baseline > gpurun_out/r4r_cfg${c}_head.json 2> gpurun_out/r4r_cfg${c}_head.err; echo "cfg$c head rc=$?"
done
python - <<'PY'
import json
for f in ("cfg1_new", "cfg1_head", "cfg2_new", "cfg2_head"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r4r_{f}.json") if l.startswith("{")][-1])
        print(f, "steps/s", round(d["value"], 2), "ms/step", round(d["ms_per_step"], 3))
    except Exception as ex:
        print(f, "ERR", ex)
PY
timeout 600 python tools/ab_flags.py --flags 0 --rounds 3 --steps 20 > gpurun_out/r4r_ab_cfg3.jsonl 2> gpurun_out/r4r_ab_cfg3.err; cat gpurun_out/r4r_ab_cfg3.jsonl
