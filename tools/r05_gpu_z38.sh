#!/bin/bash
O=gpurun_out/r05z38; mkdir -p $O
run() {
  python bench.py --mode replay --steps 80 --warmup 8 --streams $1 --host-threads $2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; h=d['host_cpu']; print('$1 streams x $2 threads [$3]: %6.0f windows/s  ms/frame %.2f; cores busy %.1f cpu us/window %.0f; assoc+ %.2f problems %.2f batch_create %.2f kernels %.2f finish %.2f' % (d['value'], d['ms_per_step'], h['cores_busy_mean'], h['cpu_us_per_window'], p['assoc+triangulate+window'], p['problems'], p['batch_create'], p['kernels'], p['finish_frames']))"
}
{
for rep in 1 2 3; do
  for cfg in "8 2" "32 2" "128 4"; do
    set -- $cfg
    run $1 $2 "problem objects recycled (default)"
    TCV_NO_PROBLEM_POOL=1 run $1 $2 "new / delete per problem (before)"
  done
done
} > $O/pool.txt 2>&1
cat $O/pool.txt
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
