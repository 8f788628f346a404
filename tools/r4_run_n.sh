#!/bin/bash
# round-4 GPU call N: s_setprio around the MFMA phases of the self-attention kernel (probe + same-box A/B of the step, debug bit 14),
# sustained MFMA rate of the box (tools/probes/mfma_peak), two concurrent half-batch chains vs one batch-7 forward
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 tools/probes/mfma_peak > gpurun_out/r4n_mfma_peak.txt 2>&1; cat gpurun_out/r4n_mfma_peak.txt
timeout 300 tools/probes/attn_bench > gpurun_out/r4n_attn_probe.txt 2>&1; cat gpurun_out/r4n_attn_probe.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or softmax" > gpurun_out/r4n_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4n_tests.log
RTDIFF_DEBUG_FLAGS=16384 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or softmax" > gpurun_out/r4n_tests_prio.log 2>&1; echo "tests(prio) rc=$?"; tail -3 gpurun_out/r4n_tests_prio.log
timeout 600 python tools/ab_flags.py --flags 0 16384 --rounds 4 --steps 20 --profile > gpurun_out/r4n_ab_prio.jsonl 2> gpurun_out/r4n_ab_prio.err; echo "ab rc=$?"; cat gpurun_out/r4n_ab_prio.jsonl; tail -2 gpurun_out/r4n_ab_prio.err
timeout 300 python tools/two_chain_probe.py > gpurun_out/r4n_two_chain.txt 2>&1; echo "two-chain rc=$?"; tail -8 gpurun_out/r4n_two_chain.txt
