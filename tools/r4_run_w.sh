#!/bin/bash
# round-4 GPU call W: kernel profile of the precise SDXL VAE guidance call on the final tree
cd "$(dirname "$0")/.."
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_vae; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o vae -- python $ROOT/tools/vae_precise_profile.py > $OUT/run.log 2> $OUT/rocprof.log); echo "rc=$?"; tail -1 $OUT/run.log
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python tools/rocpd_summary.py $DB gpurun_out/r4w_vae_precise_kernel_stats.csv | head -30
rm -rf $OUT
