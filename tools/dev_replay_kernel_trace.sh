# kernel durations of a lock-step replay (developer tool, GPU box): rocprofv3 --kernel-trace --stats of the replay bench at one host thread
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/replay_trace
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/replay_trace -o rp --output-format csv -- python3 $R/bench.py --mode replay --steps 60 --warmup 10 --host-threads ${T:-1} --no-cpu-baseline > $R/gpurun_out/replay_trace.log 2>&1
F=$(find $R/gpurun_out/replay_trace -name "*kernel_stats.csv" | head -1)
cp $F $R/gpurun_out/replay_kernel_stats.csv
cut -c1-230 $F | head -20
rm -rf $R/gpurun_out/replay_trace
