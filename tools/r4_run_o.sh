#!/bin/bash
# round-4 GPU call O: sustained MFMA rate (in-place inline-asm chains), s_setprio patterns of the self-attention kernel (probe)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 tools/probes/mfma_peak > gpurun_out/r4o_mfma_peak.txt 2>&1; cat gpurun_out/r4o_mfma_peak.txt
timeout 300 tools/probes/attn_bench > gpurun_out/r4o_attn_probe.txt 2>&1; cat gpurun_out/r4o_attn_probe.txt
