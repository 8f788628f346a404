#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/prof_store
export TMPDIR=/tmp
ROOT=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_store -o store -- python $ROOT/tools/attn_store_bench.py > $ROOT/gpurun_out/r4g_store.txt 2>&1)
find gpurun_out/prof_store -name "*kernel_stats*" | head; f=$(find gpurun_out/prof_store -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-220
python tools/rocpd_summary.py $(ls gpurun_out/prof_store/*.db gpurun_out/prof_store/*/*.db 2>/dev/null | head -1) gpurun_out/r4g_store_kernel_stats.csv | head -12
rm -rf gpurun_out/prof_store
