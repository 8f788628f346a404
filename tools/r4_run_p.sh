#!/bin/bash
# round-4 GPU call P: rocprofv3 kernel trace of the plain pass (batch-2 forward) - where does a 30 ms plain step go?
cd "$(dirname "$0")/.."
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_plain; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o plain -- python $ROOT/tools/plain_profile.py --steps 12 > $OUT/run.log 2> $OUT/rocprof.log); echo "rc=$?"; cat $OUT/run.log | tail -2
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python tools/rocpd_summary.py $DB gpurun_out/r4p_plain_kernel_stats.csv | head -40
rm -rf $OUT
