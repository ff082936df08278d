#!/bin/bash
O=gpurun_out/r05z39; mkdir -p $O
TCV_BENCH_DEBUG_POOLS=1 python bench.py --mode replay --streams 128 --steps 280 --warmup 10 --no-cpu-baseline > $O/long128.json 2> $O/long128.err; grep "\[bench\]" $O/long128.err; python3 -c "
import json; d=json.loads(open('$O/long128.json').read().strip().splitlines()[-1]); print('128 streams, 280 frames:', round(d['value']), 'windows/s', d['plan_cache'])"
TCV_BENCH_DEBUG_POOLS=1 python bench.py --mode replay --streams 8 --steps 330 --warmup 10 --no-cpu-baseline > $O/long8.json 2> $O/long8.err; grep "\[bench\]" $O/long8.err; python3 -c "
import json; d=json.loads(open('$O/long8.json').read().strip().splitlines()[-1]); print('8 streams, 330 frames:', round(d['value']), 'windows/s')"
grep -i "error\|traceback" $O/*.err | head -5
