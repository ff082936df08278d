#!/bin/bash
# round 5: lock-step replay, 8 streams: host threads x hardware queues; replay bench line with its CPU baseline; laps at 128 streams
O=gpurun_out/r05e; mkdir -p $O
for Q in 4 8; do for T in 2 3 4; do for rep in 1 2; do
  GPU_MAX_HW_QUEUES=$Q python bench.py --mode replay --steps 80 --warmup 10 --streams 8 --host-threads $T --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; print('queues $Q threads $T: %6.0f windows/s  kernels %.3f batch_create %.3f assoc+ %.3f' % (d['value'], p['kernels'], p['batch_create'], p['assoc+triangulate+window']))"
done; done; done > $O/threads_queues.txt 2>&1
( time python bench.py --mode replay --steps 100 --warmup 10 ) > $O/bench_replay.json 2> $O/bench_replay.err
TCV_DEBUG_EST=1 python bench.py --mode replay --steps 20 --warmup 10 --streams 128 --no-cpu-baseline > /dev/null 2> $O/est.err
grep "^\[est\]" $O/est.err | tail -30 > $O/est_laps.txt; rm -f $O/est.err
cat $O/threads_queues.txt; tail -5 $O/bench_replay.err
