cd $GRAFT_REPO_ROOT
./tools/probes/conv_bench > gpurun_out/r2k_conv.txt 2>&1
python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv" 2>&1 | tail -4 > gpurun_out/r2k_tests.log
for c in 1 2; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c', round(d['value'],2), round(d['ms_per_step'],3))" >> gpurun_out/r2k_sweep.txt; done
cat gpurun_out/r2k_conv.txt gpurun_out/r2k_tests.log gpurun_out/r2k_sweep.txt
