#!/bin/bash
# round-4 GPU call T: the whole GPU test suite + smoke on the tree with the grouped launches, small-batch tiles and s_setprio
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r4t_gpu_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4t_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4t_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r4t_smoke.log
