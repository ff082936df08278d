cd $GRAFT_REPO_ROOT
(for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Socket Power|Average Graphics" | tr '\n' ' '; echo; sleep 0.25; done) > gpurun_out/clk.log 2>&1 &
sleep 1
python bench.py --no-cpu-baseline --steps 600 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
wait
sed -n 1,40p gpurun_out/clk.log | cut -c1-200 | awk 'NR%3==0'
