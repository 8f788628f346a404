#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/prof_c5
export TMPDIR=/tmp
ROOT=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_c5 -o c5 -- python $ROOT/tools/bench_configs.py --configs 5 --steps 4 > $ROOT/gpurun_out/r4h_c5.txt 2>&1)
python tools/rocpd_summary.py $(ls gpurun_out/prof_c5/*.db gpurun_out/prof_c5/*/*.db 2>/dev/null | head -1) gpurun_out/r4h_config5_kernel_stats.csv | head -40
rm -rf gpurun_out/prof_c5
timeout 900 python tools/bench_configs.py > gpurun_out/r4h_configs_1_2_5.jsonl 2> gpurun_out/r4h_configs.err; cat gpurun_out/r4h_configs_1_2_5.jsonl | cut -c1-300
