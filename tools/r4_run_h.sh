#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/prof_c5
export TMPDIR=/tmp
ROOT=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_c5 -o c5 -- python $ROOT/tools/vae_precise_profile.py > $ROOT/gpurun_out/r4h_vae.txt 2>&1)
cat gpurun_out/r4h_vae.txt | tail -8
python tools/rocpd_summary.py $(ls gpurun_out/prof_c5/*.db gpurun_out/prof_c5/*/*.db 2>/dev/null | head -1) gpurun_out/r4h_vae_kernel_stats.csv | head -30
rm -rf gpurun_out/prof_c5
