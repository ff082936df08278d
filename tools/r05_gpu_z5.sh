#!/bin/bash
O=gpurun_out/r05z5; mkdir -p $O
for T in 2 4 8; do for Q in 8 16; do for rep in 1 2; do
  GPU_MAX_HW_QUEUES=$Q python bench.py --mode replay --steps 80 --warmup 10 --streams 8 --host-threads $T --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; print('8 streams, host threads $T, queues $Q: %6.0f windows/s  (ms/frame %.2f; kernels lap %.3f assoc+ %.3f batch_create %.3f)' % (d['value'], d['ms_per_step'], p['kernels'], p['assoc+triangulate+window'], p['batch_create']))"
done; done; done > $O/threads8.txt 2>&1
cat $O/threads8.txt
