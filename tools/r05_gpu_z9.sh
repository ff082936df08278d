#!/bin/bash
O=gpurun_out/r05z9; mkdir -p $O
TCV_DEBUG_EST_CPU=1 TCV_DEBUG_PACK=1 python bench.py --mode replay --steps 40 --warmup 8 --streams 64 --host-threads 1 --no-cpu-baseline > $O/one.json 2> $O/one.err
grep "est cpu" $O/one.err | tail -7
grep "batch_create\] n 64: plans" $O/one.err | tail -5
grep "batch_create\] n 64: pack" $O/one.err | tail -5
python3 -c "
import json; d=json.loads(open('$O/one.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['host_cpu'], d['native_profile_ms_per_call'])"
for T in 1 8 16; do TCV_PACK_BENCH_FRAME=64 PACK_BENCH_WINDOWS=1280 TCV_DEBUG_PACK2=1 python tools/dev_pack_bench.py $T 2>&1 | grep "pack\]\|threads"; done
cat /sys/fs/cgroup/cpu.stat | grep thrott
