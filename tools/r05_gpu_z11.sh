#!/bin/bash
O=gpurun_out/r05z11; mkdir -p $O
run() {
  python bench.py --mode replay --steps 60 --warmup 8 --streams $1 --host-threads $2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; h=d['host_cpu']; print('$1 streams x $2 threads [$3]: %6.0f windows/s  ms/frame %.2f; cores busy %.1f cpu us/window %.0f; assoc+ %.2f problems %.2f batch_create %.2f kernels %.2f' % (d['value'], d['ms_per_step'], h['cores_busy_mean'], h['cpu_us_per_window'], p['assoc+triangulate+window'], p['problems'], p['batch_create'], p['kernels']))"
}
{
for T in 8 16; do
  TCV_PACK_BENCH_FRAME=64 PACK_BENCH_WINDOWS=1280 python tools/dev_pack_bench.py $T 2>&1 | grep "threads" | sed 's/$/  [one window at a time]/'
  TCV_PACK_BENCH_STRIDED=1 TCV_PACK_BENCH_FRAME=64 PACK_BENCH_WINDOWS=1280 python tools/dev_pack_bench.py $T 2>&1 | grep "threads" | sed 's/$/  [fixed share per thread]/'
  TCV_WORKER_SPIN_US=0 TCV_PACK_BENCH_FRAME=64 PACK_BENCH_WINDOWS=1280 python tools/dev_pack_bench.py $T 2>&1 | grep "threads" | sed 's/$/  [one window at a time, workers sleep at once]/'
done
for S in 8 64 128; do for rep in 1 2 3; do
  run $S 2 "defaults: items, poll 40 us, second-sight cache"
  TCV_WORKER_SPIN_US=0 run $S 2 "items, sleep at once"
done; done
run 128 4 "defaults"
run 128 4 "defaults"
} > $O/ab.txt 2>&1
cat $O/ab.txt
