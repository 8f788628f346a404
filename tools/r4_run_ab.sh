#!/bin/bash
# round-4 GPU call AB (last seconds of the budget): the edited full-architecture test plumbing (oracle_cache wiring) on the SD-v1.5 stream-mode test
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 50 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "sd15_full_architecture_stream" > gpurun_out/r4ab_test.log 2>&1; echo "rc=$?"; grep -E "passed|failed|oracle outputs|rel-L2|Error" gpurun_out/r4ab_test.log | tail -8
