#!/bin/bash
# round-4 GPU call D: new full-size guided loop tests + two-rank CLI test, attention-store micro-benchmark, rocprof kernel stats + PMC passes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_checkpoint_gpu.py -x -q -k "guided_loop or two_ranks or cli" -s > gpurun_out/r4d_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4d_tests.log
grep -E "rel-L2|passed|failed|rc=" gpurun_out/r4d_tests.log | tail -8
timeout 300 python tools/attn_store_bench.py > gpurun_out/r4d_attn_store_bench.txt 2>&1; cat gpurun_out/r4d_attn_store_bench.txt
timeout 900 bash tools/profile_step.sh r4d --steps 10 --warmup 3 --no-extras > gpurun_out/r4d_profile.log 2>&1; echo "profile rc=$?"; tail -25 gpurun_out/r4d_profile.log
timeout 1500 bash tools/pmc_passes.sh r4d > gpurun_out/r4d_pmc.log 2>&1; echo "pmc rc=$?"; tail -5 gpurun_out/r4d_pmc.log
