#!/bin/bash
# round-4 GPU call X: precise VAE with the bf16 pair written by the convolution epilogue / the GroupNorm backward (no split kernels):
# VAE tests incl. the full-size precise decode + gradient, guidance-call timing against the round-4 HEAD library on the same box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
HEADLIB=$PWD/rich-text-to-image_amd/librtdiff_head.so
timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_fullsize_gpu.py -x -q -k "vae" -s > gpurun_out/r4x_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|flips|decode|grad" gpurun_out/r4x_tests.log | tail -12
timeout 300 python tools/vae_precise_profile.py 2>&1 | tail -1 | tee gpurun_out/r4x_vae_new.txt
RTDIFF_ALLOW_MISSING_SYMBOLS=1 RTDIFF_LIB_PATH=$HEADLIB timeout 300 python tools/vae_precise_profile.py 2>&1 | tail -1 | tee gpurun_out/r4x_vae_head.txt
timeout 300 python tools/vae_precise_profile.py 2>&1 | tail -1 | tee gpurun_out/r4x_vae_new2.txt
