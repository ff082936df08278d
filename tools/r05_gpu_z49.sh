#!/bin/bash
O=gpurun_out/r05z49; mkdir -p $O
run() {
  timeout 120 python bench.py --mode replay --steps 100 --warmup 10 --streams 8 --host-threads $1 --no-cpu-baseline 2>/dev/null < /dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; print('8 streams, host threads $1, hardware queues $2: %6.0f windows/s  ms/frame %.3f; assoc+ %.3f batch_create %.3f kernels %.3f' % (d['value'], d['ms_per_step'], p['assoc+triangulate+window'], p['batch_create'], p['kernels']))"
}
for rep in 1 2 3; do
  for T in 1 2; do for Q in 4 8; do GPU_MAX_HW_QUEUES=$Q run $T $Q; done; done
done > $O/q.txt 2>&1
cat $O/q.txt
