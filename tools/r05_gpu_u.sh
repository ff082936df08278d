#!/bin/bash
O=gpurun_out/r05u; mkdir -p $O
for S in 8 32 64 128; do for H in 2 3 4; do
  TCV_COOP_H=$H python bench.py --mode replay --steps 60 --warmup 10 --streams $S --no-cpu-baseline 2> $O/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; k=d.get('kernel_ms',{}); print('streams $S pipelined, helpers $H: %6.0f windows/s  (ms/frame %.2f; solve kernel %.3f marg %.3f; kernels lap %.3f)' % (d['value'], d['ms_per_step'], k.get('solve') or 0, k.get('marginalize') or 0, p['kernels']))" || tail -3 $O/err.txt
done; done > $O/pipe_h.txt 2>&1
cat $O/pipe_h.txt
