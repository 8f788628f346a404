cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/r2_gpu_tests.log
python __graft_entry__.py smoke > gpurun_out/r2_smoke.log 2>&1
python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench_s20.json 2>> gpurun_out/r2_bench.err
bash tools/profile_step.sh r2_final > /dev/null 2>&1
for c in 1 2 5; do python bench.py --config $c 2>/dev/null | tail -1 >> gpurun_out/r2_configs_1_2_5.jsonl.new; done
tail -5 gpurun_out/r2_gpu_tests.log; tail -2 gpurun_out/r2_smoke.log
