cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "in_situ or tile_configurations" 2>&1 | tail -3 > gpurun_out/r2r_tests.log
bash tools/pmc_passes.sh r2f > gpurun_out/r2r_pmc.log 2>&1
cat gpurun_out/r2r_tests.log; tail -3 gpurun_out/r2r_pmc.log
