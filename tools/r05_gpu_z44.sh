#!/bin/bash
O=gpurun_out/r05z44; mkdir -p $O
python -m pytest tests/test_gpu_replay.py tests/test_gpu_teacher.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
for rep in 1 2 3 4 5; do python bench.py --mode replay --steps 100 --warmup 10 --streams 8 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; print('8 streams: %6.0f windows/s  ms/frame %.3f; batch_create %.3f kernels %.3f' % (d['value'], d['ms_per_step'], p['batch_create'], p['kernels']))"; done
for rep in 1 2; do python bench.py --mode replay --steps 60 --warmup 10 --streams 128 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; print('128 streams: %6.0f windows/s  ms/frame %.3f; batch_create %.3f kernels %.3f' % (d['value'], d['ms_per_step'], p['batch_create'], p['kernels']))"; done
