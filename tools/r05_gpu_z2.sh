#!/bin/bash
O=gpurun_out/r05z2; mkdir -p $O
for S in 8 16 32; do for T in 1 2; do for Q in 4 8; do for rep in 1 2; do
  GPU_MAX_HW_QUEUES=$Q TCV_EST_MARG_AUX=1 python bench.py --mode replay --steps 80 --warmup 10 --streams $S --host-threads $T --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; print('$S streams, host threads $T, queues $Q, marginalisation on the second stream: %6.0f windows/s  (ms/frame %.2f; assoc+ %.3f)' % (d['value'], d['ms_per_step'], p['assoc+triangulate+window']))"
done; done; done; done > $O/marg_aux2.txt 2>&1
cat $O/marg_aux2.txt
