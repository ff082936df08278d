#!/bin/bash
O=gpurun_out/r05z10; mkdir -p $O
run() { # label, env..., streams, threads
  python bench.py --mode replay --steps 60 --warmup 8 --streams $1 --host-threads $2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; h=d['host_cpu']; print('$1 streams x $2 threads [$3]: %6.0f windows/s  ms/frame %.2f; cores busy %.1f cpu us/window %.0f; assoc+ %.2f problems %.2f batch_create %.2f kernels %.2f' % (d['value'], d['ms_per_step'], h['cores_busy_mean'], h['cpu_us_per_window'], p['assoc+triangulate+window'], p['problems'], p['batch_create'], p['kernels']))"
}
for S in 8 64 128; do for rep in 1 2; do
  T=2
  TCV_WORKER_SPIN_US=0 TCV_PLAN_CACHE_EAGER=1 run $S $T "sleep at once, eager cache (round-5 state)"
  TCV_WORKER_SPIN_US=40 TCV_PLAN_CACHE_EAGER=1 run $S $T "poll 40 us, eager cache"
  TCV_WORKER_SPIN_US=0 run $S $T "sleep at once, second-sight cache"
  TCV_WORKER_SPIN_US=40 run $S $T "poll 40 us, second-sight cache"
  TCV_WORKER_SPIN_US=150 run $S $T "poll 150 us, second-sight cache"
done; done > $O/ab.txt 2>&1
TCV_WORKER_SPIN_US=40 run 128 4 "poll 40 us, second-sight cache" >> $O/ab.txt 2>&1
TCV_WORKER_SPIN_US=150 run 128 4 "poll 150 us, second-sight cache" >> $O/ab.txt 2>&1
cat $O/ab.txt
for T in 1 8 16; do TCV_PACK_BENCH_FRAME=64 PACK_BENCH_WINDOWS=1280 TCV_DEBUG_PACK2=1 python tools/dev_pack_bench.py $T 2>&1 | grep "lookup\|insert\|pack_plan\|threads"; done
