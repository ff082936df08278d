#!/bin/bash
# round 5: how many worker threads the lock-step replay's host side profits from on this box (TCV_HOST_THREADS = the core grant the library assumes)
O=gpurun_out/r05n; mkdir -p $O
nproc > $O/host_threads.txt; cat /sys/fs/cgroup/cpu.max >> $O/host_threads.txt 2>/dev/null
for H in 4 8 12 16 24 32; do for rep in 1 2; do
  TCV_HOST_THREADS=$H python bench.py --mode replay --steps 50 --warmup 10 --streams 128 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; print('TCV_HOST_THREADS=$H: %6.0f windows/s  batch_create %.3f assoc+ %.3f problems %.3f finish %.3f begin %.3f' % (d['value'], p['batch_create'], p['assoc+triangulate+window'], p['problems'], p['finish_frames'], p['begin_frames']))"
done; done >> $O/host_threads.txt 2>&1
cat $O/host_threads.txt
