cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_gpu_td.py -x -q -s 2>&1 | grep -v "^$" | tail -8
cat > /tmp/tdbench.py <<'PY'
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, 'tc-viml_amd')
import synth, tcv as gpu
ws = [synth.with_time_offset(synth.window_at(synth.make_windows(100 + k, 1), 0), 100 + k) for k in range(8)]
for variant in (0, 1):
    gpu.check(gpu.lib().tcv_set_solver_variant(variant))
    Ws = [gpu.Window(ws[k % 8]) for k in range(1024)]
    b = gpu.Batch(Ws)
    o = gpu.default_options(8, True)
    for r in range(3): b.solve(o); b.synchronize()
    t = time.perf_counter()
    for r in range(10): b.solve(o)
    b.synchronize()
    print(b.plan_stats()["layout"], "1024 ESTIMATE_TD windows: %.3f ms per solve launch" % ((time.perf_counter() - t) * 100), b.plan_stats())
gpu.check(gpu.lib().tcv_set_solver_variant(0))
PY
timeout 300 python3 /tmp/tdbench.py 2>&1 | tail -4
for rep in 1 2; do
for v in x base; do
  L=tc-viml_amd/libtcv_hip_$v.so; [ $v = x ] && L=tc-viml_amd/libtcv_hip.so
  TCV_LIB=$L python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['kernel_ms'], round(d['value']))"
done; done
