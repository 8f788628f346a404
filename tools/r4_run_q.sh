#!/bin/bash
# round-4 GPU call Q: 64-row class-B tiles (variant 9) against 128-row ones on the small-batch shapes (plain pass, SD-v1.5)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 tools/probes/gemm16_bench s > gpurun_out/r4q_gemm16_smallm.txt 2>&1; cat gpurun_out/r4q_gemm16_smallm.txt
