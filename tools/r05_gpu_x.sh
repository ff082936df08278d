#!/bin/bash
O=gpurun_out/r05x; mkdir -p $O
for P in "" "TCV_BENCH_NO_PIPELINE=1"; do
  env $P python bench.py --mode replay --steps 40 --warmup 10 --streams 32 --host-threads 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$P', round(d['value']), d['ms_per_step'], d['native_profile_ms_per_call'])"
done > $O/laps.txt 2>&1
cat $O/laps.txt
