#!/bin/bash
O=gpurun_out/r05q; mkdir -p $O
for S in 8 32 64 128; do for rep in 1 2; do
  python bench.py --mode replay --steps 50 --warmup 10 --streams $S --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; k=d.get('kernel_ms',{}); print('streams $S (new helper rule): %6.0f windows/s  kernels lap %.3f  solve kernel %.3f ms marg %.3f ms' % (d['value'], p['kernels'], k.get('solve') or 0, k.get('marginalize') or 0))"
done; done > $O/coop_rule.txt 2>&1
python -m pytest tests/test_gpu_coop.py tests/test_gpu_replay.py -x -q -k "not full_length" 2>&1 | tail -3 >> $O/coop_rule.txt
cat $O/coop_rule.txt
