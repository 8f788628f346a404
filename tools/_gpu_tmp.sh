cd $GRAFT_REPO_ROOT
(for i in $(seq 1 60); do rocm-smi --showpower --showclocks --showuse --json 2>/dev/null | head -c 1500; echo; sleep 0.5; done) > gpurun_out/r2q_smi.txt &
SMI=$!
python bench.py --no-cpu-baseline > gpurun_out/r2q_bench.json 2> gpurun_out/r2q_bench.err
kill $SMI 2>/dev/null
rocm-smi --showpower --showclocks 2>&1 | head -30 > gpurun_out/r2q_smi_idle.txt
tail -3 gpurun_out/r2q_smi.txt
