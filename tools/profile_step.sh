#!/bin/bash
# rocprofv3 kernel trace of a short bench.py run + steady-state per-kernel summary.
#   tools/profile_step.sh <tag> [bench.py args...]      (run from the repo root on the GPU box; output under gpurun_out/)
set -e
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o $TAG -- python $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/rocprof.log) || { tail -5 $OUT/rocprof.log; exit 1; }
DB=$(ls $OUT/*.db | head -1)
python $ROOT/tools/rocpd_summary.py $DB $ROOT/gpurun_out/${TAG}_kernel_stats.csv > /dev/null
python $ROOT/tools/gap_analysis.py $DB 0.45 0.7 > $ROOT/gpurun_out/${TAG}_steady_state_kernels.txt
cat $ROOT/gpurun_out/${TAG}_steady_state_kernels.txt
cp $OUT/bench.json $ROOT/gpurun_out/${TAG}_bench.json; rm -rf $OUT      # the rocpd database is tens of MB: keep the summaries only
