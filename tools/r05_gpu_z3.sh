#!/bin/bash
O=gpurun_out/r05z3; mkdir -p $O
python -m pytest tests/test_gpu_replay.py tests/test_gpu_teacher.py -x -q -k "not full_length" 2>&1 | tail -3 > $O/defaults.txt
for S in 8 16 32 64 128; do for rep in 1 2; do
  python bench.py --mode replay --steps 80 --warmup 10 --streams $S --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; print('$S streams (defaults: 2 host threads, 8 hardware queues, small frames marginalise on the second stream): %6.0f windows/s  (ms/frame %.2f; assoc+ %.3f)' % (d['value'], d['ms_per_step'], p['assoc+triangulate+window']))"
done; done >> $O/defaults.txt 2>&1
python bench.py 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default line:', round(d['value']), d['kernel_ms'], 'replay', round(d['replay_windows_per_s']), 'stream', round(d['stream_solves_per_s']), d['single_window_ms']['host_blocks_to_states'])" >> $O/defaults.txt 2>&1
cat $O/defaults.txt
