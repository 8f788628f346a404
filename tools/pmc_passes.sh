#!/bin/bash
# rocprofv3 PMC passes of a short bench.py run (kernel trace only, one counter group per pass: the pool refuses --pmc together
# with sys/hip/hsa tracing).   tools/pmc_passes.sh <tag>   -> gpurun_out/<tag>_pmc_traffic.json, <tag>_pmc_sq.json
set -e
TAG=$1
ROOT=$(pwd)
export TMPDIR=/tmp
run() { # name, counters...
    local name=$1; shift
    local out=$ROOT/gpurun_out/pmc_${TAG}_$name
    mkdir -p $out
    (cd /tmp && rocprofv3 --kernel-trace --pmc "$@" -d $out -o $name -- python $ROOT/bench.py --roofline-only --warmup 2 > $out/bench.json 2> $out/rocprof.log) || { tail -5 $out/rocprof.log; exit 1; }
    ls $out/*.db | head -1
}
F=$(run fetch FETCH_SIZE)
W=$(run write WRITE_SIZE)
python $ROOT/tools/pmc_traffic.py $F $W $ROOT/gpurun_out/${TAG}_pmc_traffic.json $ROOT/gpurun_out/pmc_${TAG}_fetch/bench.json
S=$(run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE)
python $ROOT/tools/pmc_sq.py $S $ROOT/gpurun_out/${TAG}_pmc_sq.json
rm -rf $ROOT/gpurun_out/pmc_${TAG}_*      # databases are tens of MB each: keep the summaries only
