#!/bin/bash
O=gpurun_out/r05z36; mkdir -p $O
run() {
  python bench.py --mode replay --steps 80 --warmup 8 --streams $1 --host-threads $2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; h=d['host_cpu']; print('$1 streams x $2 threads [$3]: %6.0f windows/s  ms/frame %.2f; cores busy %.1f; assoc+ %.2f problems %.2f batch_create %.2f kernels %.2f' % (d['value'], d['ms_per_step'], h['cores_busy_mean'], p['assoc+triangulate+window'], p['problems'], p['batch_create'], p['kernels']))"
}
{
for rep in 1 2 3 4; do
  run 8 2 "per-estimator sections on the caller up to 8 per call (new default)"
  TCV_EST_SERIAL_MAX=0 run 8 2 "always on the worker pool (before)"
done
for S in 16 32; do for rep in 1 2; do
  run $S 2 "new default"
  TCV_EST_SERIAL_MAX=0 run $S 2 "always on the worker pool"
  TCV_EST_SERIAL_MAX=16 run $S 2 "caller up to 16 per call"
done; done
} > $O/serial2.txt 2>&1
cat $O/serial2.txt
python -m pytest tests/test_gpu_replay.py tests/test_gpu_teacher.py -m gpu -x -q 2>&1 | tail -2
