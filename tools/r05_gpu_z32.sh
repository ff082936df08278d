#!/bin/bash
O=gpurun_out/r05z32; mkdir -p $O
run() {
  python bench.py --mode replay --steps 60 --warmup 8 --streams $1 --host-threads $2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; print('$1 streams x $2 threads [$3]: %6.0f windows/s  ms/frame %.2f; assoc+ %.2f batch_create %.2f kernels %.2f' % (d['value'], d['ms_per_step'], p['assoc+triangulate+window'], p['batch_create'], p['kernels']))"
}
{
for rep in 1 2 3; do
  for S in 32 64 128; do
    T=2; [ $S = 128 ] && T=4
    run $S $T "deferred launch from 12 windows per call (default)"
    TCV_EST_MARG_DEFER=0 run $S $T "never deferred: on the second stream behind the states"
  done
done
} > $O/defer.txt 2>&1
cat $O/defer.txt
