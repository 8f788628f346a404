cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
python bench.py --no-cpu-baseline > gpurun_out/r2t_new$i.json 2> gpurun_out/r2t.err
(cd _head && python bench.py --no-cpu-baseline > ../gpurun_out/r2t_head$i.json 2>> ../gpurun_out/r2t.err)
done
python - <<'PY'
import json
for f in ['r2t_new1','r2t_head1','r2t_new2','r2t_head2','r2t_new3','r2t_head3']:
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],3), round(d['roofline']['frac'],4), {k:round(v['total_ms'],2) for k,v in d['roofline']['per_kernel'].items()})
PY
