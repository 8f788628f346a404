#!/bin/bash
# round-4 GPU call L: epilogue store policy A/B (write-back / sc1 / sc0 sc1 / nt, debug bits 10-11), batch scaling of one forward
# (7 / 14 streams in M), gemm16 parity subset on the buffer-store epilogue
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm16 or cross_attn or conv3x3_on" > gpurun_out/r4l_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4l_tests.log
timeout 600 python tools/ab_flags.py --flags 0 1024 2048 3072 --rounds 3 --steps 20 --profile > gpurun_out/r4l_ab_store.jsonl 2> gpurun_out/r4l_ab_store.err; echo "ab rc=$?"; cat gpurun_out/r4l_ab_store.jsonl; tail -3 gpurun_out/r4l_ab_store.err
timeout 300 python tools/batch_scaling.py > gpurun_out/r4l_batch_scaling.txt 2>&1; echo "scaling rc=$?"; cat gpurun_out/r4l_batch_scaling.txt | tail -6
