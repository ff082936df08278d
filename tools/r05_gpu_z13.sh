#!/bin/bash
O=gpurun_out/r05z13; mkdir -p $O
TCV_HOST_THREADS=1 TCV_DEBUG_EST_CPU=1 TCV_DEBUG_PACK=1 python bench.py --mode replay --steps 30 --warmup 8 --streams 64 --host-threads 1 --no-cpu-baseline > $O/one.json 2> $O/one.err
grep "est cpu" $O/one.err | tail -7
grep "batch_create\] n 64: plans" $O/one.err | tail -3
grep "batch_create\] n 64: pack" $O/one.err | tail -3
python3 -c "
import json; d=json.loads(open('$O/one.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['host_cpu'], d['native_profile_ms_per_call'])"
