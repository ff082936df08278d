#!/bin/bash
O=gpurun_out/r05t; mkdir -p $O
for S in 32 128; do for P in "" "TCV_BENCH_NO_PIPELINE=1"; do
  env $P TCV_BENCH_DEBUG_POOLS=1 python bench.py --mode replay --steps 60 --warmup 10 --streams $S --no-cpu-baseline 2> $O/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; print('streams $S $P: %6.0f windows/s  (ms/frame %.2f; batch_create %.3f assoc+ %.3f kernels lap %.3f)' % (d['value'], d['ms_per_step'], p['batch_create'], p['assoc+triangulate+window'], p['kernels']))"
  grep "\[bench\]" $O/err.txt
done; done > $O/pools.txt 2>&1
cat $O/pools.txt
