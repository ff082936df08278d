#!/bin/bash
# round 5: deferred marginalisation launch + native window tap: replay / teacher suites, thread sweep, spill A/B of the marginalisation kernel, chained-prior invariants
O=gpurun_out/r05f; mkdir -p $O
python -m pytest tests/test_gpu_replay.py tests/test_gpu_teacher.py tests/test_gpu_resident.py -x -q -s -k "not full_length" > $O/pytest_replay.log 2>&1; echo "pytest rc $?" >> $O/pytest_replay.log
for T in 2 4; do for E in "" "TCV_EST_MARG_EAGER=1"; do for rep in 1 2; do
  env $E python bench.py --mode replay --steps 80 --warmup 10 --streams 8 --host-threads $T --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; print('threads $T $E: %6.0f windows/s  kernels %.3f batch_create %.3f assoc+ %.3f downloads %.3f' % (d['value'], p['kernels'], p['batch_create'], p['assoc+triangulate+window'], p['downloads']))"
done; done; done > $O/defer_ab.txt 2>&1
for E in "" "TCV_EST_MARG_EAGER=1"; do
  env $E python bench.py --mode replay --steps 40 --warmup 10 --streams 128 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; print('128 streams $E: %6.0f windows/s  %s' % (d['value'], p))"
done >> $O/defer_ab.txt 2>&1
bash tools/r05_marg_spill_ab.sh > $O/marg_spill_ab.txt 2>&1
python tests/dev/chained_prior_invariants.py > $O/chained_prior_invariants.txt 2>&1
tail -5 $O/pytest_replay.log; cat $O/defer_ab.txt $O/marg_spill_ab.txt $O/chained_prior_invariants.txt
