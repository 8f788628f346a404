#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_vae_gpu.py tests/test_fullsize_gpu.py tests/test_facade_gpu.py -q -k "vae or guided or colour" -s > gpurun_out/r4i_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4i_tests.log
grep -E "rel-L2|passed|failed|rc=|guidance call" gpurun_out/r4i_tests.log | tail -14
timeout 600 python tools/bench_configs.py --configs 5,2 > gpurun_out/r4i_configs.jsonl 2> gpurun_out/r4i_configs.err; cut -c1-220 gpurun_out/r4i_configs.jsonl
RTDIFF_DEBUG_FLAGS=128 timeout 600 python tools/bench_configs.py --configs 5,2 > gpurun_out/r4i_configs_three_launches.jsonl 2> gpurun_out/r4i_configs3.err; cut -c1-220 gpurun_out/r4i_configs_three_launches.jsonl
