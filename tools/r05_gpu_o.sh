#!/bin/bash
# round 5: 128 streams against the number of host threads (each advancing its share in lock step), hardware queues and the library's core grant
O=gpurun_out/r05o; mkdir -p $O
for T in 2 4 8; do for Q in 4 8; do for H in 16 32; do
  GPU_MAX_HW_QUEUES=$Q TCV_HOST_THREADS=$H python bench.py --mode replay --steps 50 --warmup 10 --streams 128 --host-threads $T --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; print('host threads $T queues $Q grant $H: %6.0f windows/s  kernels %.3f batch_create %.3f assoc+ %.3f' % (d['value'], p['kernels'], p['batch_create'], p['assoc+triangulate+window']))"
done; done; done > $O/threads128.txt 2>&1
cat $O/threads128.txt
