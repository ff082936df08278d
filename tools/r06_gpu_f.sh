#!/bin/bash
# round 6, GPU call f: at high stream counts the device is the bound and the cooperative mode spends ~4 CUs x 1.3 - 1.8 ms per window where the one-workgroup
# kernel spends one CU x 2.5 ms: A/B of the cooperative mode against TCV_COOP_H=0 / 1 for 32 / 64 / 128 streams on 2 / 4 host threads; the round's new tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06f; mkdir -p $O; cd $R
(python -m pytest tests/test_gpu_replay.py -x -q -k "shipped_solver_budget or failed_deferred" 2>&1 | tail -5) > $O/tests_new.txt
(python -m pytest tests/test_gpu_marg.py -x -q -s -k "golden or second_new or shapes or keeps_nothing" 2>&1 | grep -E "passed|failed|J0|measured|MARGIN" | tail -12) > $O/tests_marg.txt
for S in 32 64 128; do
  for T in 2 4; do
    for spec in "default:" "no helpers:TCV_COOP_H=0" "one helper:TCV_COOP_H=1"; do
      name="${spec%%:*}"; var="${spec#*:}"
      env $var python3 bench.py --mode replay --streams $S --host-threads $T --steps 40 --warmup 8 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%3d streams, %d host threads, %-12s %8.0f windows/s   kernels per frame: %s' % ($S, $T, '$name', d['value'], {k: round(v, 3) for k, v in (d.get('kernel_ms_per_call') or {}).items()}))"
    done
  done
done > $O/replay_coop_ab.txt 2>&1
cat $O/tests_new.txt $O/tests_marg.txt $O/replay_coop_ab.txt
