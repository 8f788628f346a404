cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -q -m gpu 2>&1 | tail -4 > gpurun_out/r2j_tests.log
for ring in 4 2 3; do for tgt in 256 512; do
  RT_SPLITK_RING=$ring RT_SPLITK_TARGET=$tgt python bench.py --config 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c1 ring $ring target $tgt', round(d['value'],2), round(d['ms_per_step'],3))" >> gpurun_out/r2j_sweep.txt
done; done
for tgt in 256 512; do
  RT_SPLITK_TARGET=$tgt python bench.py --config 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 ring 4 target $tgt', round(d['value'],2), round(d['ms_per_step'],3))" >> gpurun_out/r2j_sweep.txt
done
python bench.py --no-cpu-baseline > gpurun_out/r2j_new.json 2> gpurun_out/r2j_new.err
(cd _head && python bench.py --no-cpu-baseline > ../gpurun_out/r2j_head.json 2> ../gpurun_out/r2j_head.err)
python bench.py --no-cpu-baseline > gpurun_out/r2j_new2.json 2>> gpurun_out/r2j_new.err
cat gpurun_out/r2j_tests.log gpurun_out/r2j_sweep.txt
python - <<'PY'
import json
for f in ['r2j_new','r2j_head','r2j_new2']:
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['roofline']['per_kernel']['attn_kernel<self>']['tflops'])
PY
