cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -q -m gpu 2>&1 | tail -8 > gpurun_out/r2i_tests.log
./tools/probes/attn_bench > gpurun_out/r2i_attn.txt 2>&1
python bench.py --no-cpu-baseline > gpurun_out/r2i_new.json 2> gpurun_out/r2i_new.err
(cd _head && python bench.py --no-cpu-baseline > ../gpurun_out/r2i_head.json 2> ../gpurun_out/r2i_head.err)
python bench.py --no-cpu-baseline > gpurun_out/r2i_new2.json 2>> gpurun_out/r2i_new.err
(cd _head && python bench.py --no-cpu-baseline > ../gpurun_out/r2i_head2.json 2>> ../gpurun_out/r2i_head.err)
bash tools/profile_step.sh r2i --steps 20 > /dev/null 2>&1
bash tools/profile_step.sh r2i_c1 --config 1 > /dev/null 2>&1
cat gpurun_out/r2i_tests.log gpurun_out/r2i_attn.txt
