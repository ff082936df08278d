#!/bin/bash
O=gpurun_out/r05z; mkdir -p $O
for T in 1 2; do for A in "" "TCV_EST_MARG_AUX=1"; do for rep in 1 2 3; do
  env $A python bench.py --mode replay --steps 80 --warmup 10 --streams 8 --host-threads $T --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; print('8 streams, host threads $T $A: %6.0f windows/s  (ms/frame %.2f; assoc+ %.3f kernels lap %.3f batch_create %.3f)' % (d['value'], d['ms_per_step'], p['assoc+triangulate+window'], p['kernels'], p['batch_create']))"
done; done; done > $O/marg_aux.txt 2>&1
for A in "" "TCV_EST_MARG_AUX=1"; do
  env $A python bench.py --mode replay --steps 60 --warmup 10 --streams 16 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; print('16 streams, host threads 2 $A: %6.0f windows/s  (ms/frame %.2f; assoc+ %.3f)' % (d['value'], d['ms_per_step'], p['assoc+triangulate+window']))"
done >> $O/marg_aux.txt 2>&1
cat $O/marg_aux.txt
