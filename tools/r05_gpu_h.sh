#!/bin/bash
O=gpurun_out/r05h; mkdir -p $O
python -m pytest tests/test_gpu_replay.py tests/test_gpu_lines.py tests/test_gpu_teacher.py -x -q -k "not full_length" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for S in 8 128; do for M in "" "TCV_EST_HOST_MAP=1"; do for rep in 1 2; do
  env $M python bench.py --mode replay --steps 60 --warmup 10 --streams $S --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; print('streams $S $M: %6.0f windows/s  kernels %.3f batch_create %.3f assoc+ %.3f downloads %.3f problems %.3f' % (d['value'], p['kernels'], p['batch_create'], p['assoc+triangulate+window'], p['downloads'], p['problems']))"
done; done; done > $O/map_ab.txt 2>&1
tail -3 $O/pytest.log; cat $O/map_ab.txt
