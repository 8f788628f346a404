#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r4_gpu_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4_gpu_tests.log
tail -5 gpurun_out/r4_gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r4_smoke.log
