#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_fullsize_gpu.py tests/test_checkpoint_gpu.py -q -k "guided_loop or two_ranks or cli" -s > gpurun_out/r4e_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4e_tests.log
grep -E "rel-L2|passed|failed|rc=|Error|error" gpurun_out/r4e_tests.log | tail -12
