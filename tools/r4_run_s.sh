#!/bin/bash
# round-4 GPU call S: grouped Q|K + V^T launches of the plain pass (pairs 2 / 3), fused cross-attention on / off (debug bit 4) in the
# plain pass and in the rich-text step, same box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "grouped" > gpurun_out/r4s_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4s_tests.log
for f in 0 8192 16 0; do
  echo -n "plain pass, RTDIFF_DEBUG_FLAGS=$f: "; RTDIFF_DEBUG_FLAGS=$f timeout 300 python tools/plain_profile.py --steps 20 2>&1 | tail -1
done | tee gpurun_out/r4s_plain_ab.txt
timeout 600 python tools/ab_flags.py --flags 0 16 --rounds 4 --steps 20 > gpurun_out/r4s_ab_xattn.jsonl 2> gpurun_out/r4s_ab_xattn.err; cat gpurun_out/r4s_ab_xattn.jsonl
