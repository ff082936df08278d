#!/bin/bash
# round 5: every host thread alternating between two halves of its streams (host side of one against the kernels of the other) vs one lock-step object per thread
O=gpurun_out/r05r; mkdir -p $O
python -m pytest tests/test_gpu_replay.py tests/test_gpu_teacher.py -x -q -k "not full_length" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for S in 8 32 64 128; do for P in "TCV_BENCH_PIPELINE=1" ""; do for rep in 1 2; do
  env $P python bench.py --mode replay --steps 60 --warmup 10 --streams $S --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; k=d.get('kernel_ms',{}); print('streams $S $P: %6.0f windows/s  (ms/frame %.2f; solve kernel %.3f marg %.3f; batch_create %.3f assoc+ %.3f)' % (d['value'], d['ms_per_step'], k.get('solve') or 0, k.get('marginalize') or 0, p['batch_create'], p['assoc+triangulate+window']))"
done; done; done > $O/pipeline.txt 2>&1
cat $O/pipeline.txt
