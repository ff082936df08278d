#!/bin/bash
O=gpurun_out/r05z22; mkdir -p $O
run() {
  python bench.py --mode replay --steps 60 --warmup 8 --streams $1 --host-threads $2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; h=d['host_cpu']; print('$1 streams x $2 threads [$3]: %6.0f windows/s  ms/frame %.2f; cores busy %.1f; batch_create %.2f kernels %.2f solve kernel %.2f' % (d['value'], d['ms_per_step'], h['cores_busy_mean'], p['batch_create'], p['kernels'], d['kernel_ms']['solve']))"
}
{
for rep in 1 2; do
for H in auto 0 2 3 5; do
  if [ $H = auto ]; then run 128 4 "helpers: library rule"; else TCV_COOP_H=$H run 128 4 "helpers $H"; fi
done
for H in auto 0 2 3; do
  if [ $H = auto ]; then run 128 2 "helpers: library rule"; else TCV_COOP_H=$H run 128 2 "helpers $H"; fi
done
done
} > $O/helpers.txt 2>&1
cat $O/helpers.txt
