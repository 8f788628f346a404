#!/bin/bash
# round-4 GPU call M: grouped Q|K + V^T launch - parity (bit-identical with two launches) and same-box A/B of the step (debug bit 13),
# plus the round-4 HEAD library (librtdiff_head.so, built from commit c5e9ad9) on the same box as the reference point
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "grouped or gemm16 or cross_attn" > gpurun_out/r4m_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4m_tests.log
timeout 600 python tools/ab_flags.py --flags 0 8192 --rounds 4 --steps 20 --profile > gpurun_out/r4m_ab_pair.jsonl 2> gpurun_out/r4m_ab_pair.err; echo "ab rc=$?"; cat gpurun_out/r4m_ab_pair.jsonl; tail -2 gpurun_out/r4m_ab_pair.err
RTDIFF_ALLOW_MISSING_SYMBOLS=1 RTDIFF_LIB_PATH=$PWD/rich-text-to-image_amd/librtdiff_head.so timeout 600 python tools/ab_flags.py --flags 0 --rounds 4 --steps 20 --profile > gpurun_out/r4m_ab_head.jsonl 2> gpurun_out/r4m_ab_head.err; echo "head rc=$?"; cat gpurun_out/r4m_ab_head.jsonl; tail -2 gpurun_out/r4m_ab_head.err
