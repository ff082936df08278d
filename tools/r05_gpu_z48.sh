#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05z48; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/st8 -o st --output-format csv -- python3 $R/bench.py --mode replay --streams 8 --steps 40 --warmup 10 --host-threads 2 --no-cpu-baseline > $O/b8.json 2> /dev/null < /dev/null
F=$(find $O/st8 -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && grep "lines_match\|lines_fov\|solve_kernel" "$F" | cut -c1-120
timeout 200 rocprofv3 --kernel-trace --stats -d $O/st128 -o st --output-format csv -- python3 $R/bench.py --mode replay --streams 128 --steps 30 --warmup 8 --no-cpu-baseline > $O/b128.json 2> /dev/null < /dev/null
F=$(find $O/st128 -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && grep "lines_match\|lines_fov\|solve_kernel" "$F" | cut -c1-120
rm -rf $O/st8 $O/st128
cd $R
for rep in 1 2 3; do timeout 120 python bench.py --mode replay --steps 100 --warmup 10 --streams 8 --no-cpu-baseline 2>/dev/null < /dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; print('8 streams: %6.0f windows/s  ms/frame %.3f; assoc+ %.3f kernels %.3f' % (d['value'], d['ms_per_step'], p['assoc+triangulate+window'], p['kernels']))"; done
