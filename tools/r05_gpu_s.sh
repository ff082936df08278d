#!/bin/bash
O=gpurun_out/r05s; mkdir -p $O
python bench.py --mode replay --steps 30 --warmup 10 --streams 8 --no-cpu-baseline > $O/p8.json 2> $O/p8.err; tail -30 $O/p8.err | cut -c1-300 > $O/p8_tail.txt
for Q in 8 16; do for S in 32 128; do
  GPU_MAX_HW_QUEUES=$Q python bench.py --mode replay --steps 60 --warmup 10 --streams $S --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; k=d.get('kernel_ms',{}); print('queues $Q streams $S pipelined: %6.0f windows/s  (ms/frame %.2f; solve kernel %.3f marg %.3f; batch_create %.3f assoc+ %.3f kernels lap %.3f)' % (d['value'], d['ms_per_step'], k.get('solve') or 0, k.get('marginalize') or 0, p['batch_create'], p['assoc+triangulate+window'], p['kernels']))"
done; done > $O/pipeline_q.txt 2>&1
cat $O/p8_tail.txt $O/pipeline_q.txt
