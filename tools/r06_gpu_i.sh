#!/bin/bash
# round 6, GPU call i: the gauge fix in the solve kernel's epilogue -- bit identity tests, A/B of the benchmark step and of the 8-stream replay
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06i; mkdir -p $O; cd $R
(python -m pytest tests/test_gpu_gauge.py tests/test_gpu_coop.py tests/test_gpu_replay.py tests/test_gpu_teacher.py -x -q 2>&1 | tail -6) > $O/tests.txt
for rep in 1 2 3; do
  for spec in "fused:" "separate kernel:TCV_BENCH_SEPARATE_GAUGE=1"; do
    name="${spec%%:*}"; var="${spec#*:}"
    env $var python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-18s step %.4f ms  solve %.3f ms  marginalisation %.3f ms  %.1f K solves/s' % ('$name', d['ms_per_step'], d['kernel_ms']['solve'], d['kernel_ms']['marginalize'], d['value'] / 1e3))"
  done
done > $O/gauge_ab.txt 2>&1
for rep in 1 2; do
  for spec in "fused:" "separate kernel:TCV_EST_SEPARATE_GAUGE=1"; do
    name="${spec%%:*}"; var="${spec#*:}"
    env $var python3 bench.py --mode replay --steps 100 --warmup 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-18s 8-stream replay %.0f windows/s' % ('$name', d['value']))"
  done
done >> $O/gauge_ab.txt 2>&1
python3 tools/dev_single_latency.py 2>&1 | tail -4 >> $O/gauge_ab.txt
cat $O/tests.txt $O/gauge_ab.txt
