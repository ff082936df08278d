#!/bin/bash
# rocprofv3 --kernel-trace --stats of the PLAIN default bench command (extras included: stream mode on four host threads, replay on two)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05z47; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/st -o st --output-format csv -- python3 $R/bench.py > $O/bench_under_rocprof.json 2> $O/err.txt
echo "rc $?"
tail -c 300 $O/bench_under_rocprof.json | head -c 200; echo
F=$(find $O/st -name "*kernel_stats.csv" | head -1); cp $F $O/kernel_stats_default_command.csv; cut -c1-160 $F | head -12
