#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_vae_gpu.py tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -q -k "vae or patch_kernel" -s > gpurun_out/r4i_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4i_tests.log
grep -E "precise|passed|failed|rc=|guidance call|full-width" gpurun_out/r4i_tests.log | tail -12
timeout 600 python tools/vae_bench.py 2>&1 | tail -7
timeout 600 python tools/bench_configs.py --configs 5,2 > gpurun_out/r4i_configs.jsonl 2> gpurun_out/r4i_configs.err; python -c "
import json
for l in open('gpurun_out/r4i_configs.jsonl'):
    d=json.loads(l); print(d['config'], round(d['value'],2), 'steps/s')"
