#!/bin/bash
# round-4 GPU call Y: dense hi / lo contractions of the precise VAE as one launch on the gemm16 fp32 kernels (debug bit 15 = three launches)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_fullsize_gpu.py -x -q -k "vae" -s > gpurun_out/r4y_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|precise" gpurun_out/r4y_tests.log | tail -8
for f in 0 32768 0 32768; do echo -n "flags $f: "; RTDIFF_DEBUG_FLAGS=$f timeout 300 python tools/vae_precise_profile.py 2>&1 | tail -1; done | tee gpurun_out/r4y_vae_ab.txt
