#!/bin/bash
O=gpurun_out/r05z7; mkdir -p $O
for S in 64 128; do for T in 2 3 4 6; do for Q in 8 16; do
  GPU_MAX_HW_QUEUES=$Q python bench.py --mode replay --steps 50 --warmup 8 --streams $S --host-threads $T --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; tot=sum(v for k,v in p.items() if k!='calls'); print('$S streams, host threads $T, queues $Q: %6.0f windows/s  (ms/frame %.2f; native sum %.2f;' % (d['value'], d['ms_per_step'], tot), ' '.join('%s %.2f' % (k[:8],v) for k,v in p.items() if k!='calls'), ')')"
done; done; done > $O/threads.txt 2>&1
cat $O/threads.txt
