#!/bin/bash
# round 5: cooperative helpers per window against the number of streams: a 64-window call fills the chip with 1 + 3 workgroups per window, two host
# threads' calls then queue behind each other -- is the plain one-workgroup-per-window launch (TCV_COOP_H=0) better from some batch size on?
O=gpurun_out/r05p; mkdir -p $O
for S in 32 64 128; do for H in auto 0 1 2; do
  if [ $H = auto ]; then E=""; else E="TCV_COOP_H=$H"; fi
  env $E python bench.py --mode replay --steps 50 --warmup 10 --streams $S --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; k=d.get('kernel_ms',{}); print('streams $S helpers $H: %6.0f windows/s  kernels lap %.3f  solve kernel %.3f ms marg %.3f ms' % (d['value'], p['kernels'], k.get('solve') or 0, k.get('marginalize') or 0))"
done; done > $O/coop_h.txt 2>&1
cat $O/coop_h.txt
