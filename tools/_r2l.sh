cd $GRAFT_REPO_ROOT
./tools/probes/conv_bench 2>&1 | tail -6 > gpurun_out/r2l_conv.txt
python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv" 2>&1 | tail -4 > gpurun_out/r2l_tests.log
python bench.py --no-cpu-baseline > gpurun_out/r2l_new.json 2> gpurun_out/r2l_new.err
(cd _head && python bench.py --no-cpu-baseline > ../gpurun_out/r2l_head.json 2> ../gpurun_out/r2l_head.err)
python bench.py --no-cpu-baseline > gpurun_out/r2l_new2.json 2>> gpurun_out/r2l_new.err
(cd _head && python bench.py --no-cpu-baseline > ../gpurun_out/r2l_head2.json 2>> ../gpurun_out/r2l_head.err)
for c in 1 2 5; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c', round(d['value'],2), round(d['ms_per_step'],3))" >> gpurun_out/r2l_sweep.txt; done
cat gpurun_out/r2l_conv.txt gpurun_out/r2l_tests.log gpurun_out/r2l_sweep.txt
python - <<'PY'
import json
for f in ['r2l_new','r2l_head','r2l_new2','r2l_head2']:
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], {k:round(v['total_ms'],2) for k,v in d['roofline']['per_kernel'].items()})
PY
