#!/bin/bash
O=gpurun_out/r05z4; mkdir -p $O
for Q in 4 8; do for rep in 1 2; do
  GPU_MAX_HW_QUEUES=$Q python bench.py --mode stream --windows 2048 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $Q: stream mode %6.0f solves/s (host priors %6.0f)' % (d['stream_solves_per_s'], d['stream_host_priors_solves_per_s']))"
done; done > $O/stream_q.txt 2>&1
for Q in 4 8; do
  GPU_MAX_HW_QUEUES=$Q python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $Q: default line: stream %6.0f replay %6.0f' % (d['stream_solves_per_s'], d['replay_windows_per_s']))"
done >> $O/stream_q.txt 2>&1
cat $O/stream_q.txt
