#!/bin/bash
# round 6, GPU call e: per-part timers of the four-wavefront register tridiagonalisation, kernel durations of a replay frame (pre-integration kernel
# after its re-cut), the default bench line with its extras
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06e; mkdir -p $O; cd $R
for F in 4 0; do
  echo "== TCV_MARG_EIG_FLAGS=$F, 256 threads, one workgroup per CU (8 windows)" >> $O/phase_cycles_marg.txt
  TCV_MARG_EIG_FLAGS=$F TCV_MARG_NT=256 TCV_LIB=tc-viml_amd/libtcv_hip_prof.so TCV_DEBUG=1 python3 tools/dev_marg_profile.py 2>&1 | grep -E "window 0|eig_rr.tridiag|tridiag steps|tridiag_cols4" | head -5 >> $O/phase_cycles_marg.txt
done
python3 bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/rp8 -o r --output-format csv -- python3 $R/bench.py --mode replay --steps 60 --warmup 10 > $O/rp8.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/rp128 -o r --output-format csv -- python3 $R/bench.py --mode replay --streams 128 --steps 30 --warmup 8 > $O/rp128.log 2>&1
cd $R
find $O/rp8 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/replay8_kernel_stats.csv
find $O/rp128 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/replay128_kernel_stats.csv
rm -rf $O/rp8 $O/rp128
cat $O/phase_cycles_marg.txt; head -8 $O/replay8_kernel_stats.csv | cut -c1-150; head -8 $O/replay128_kernel_stats.csv | cut -c1-150
python3 -c "
import json; d=json.load(open('$O/bench.json'))
print({k: d[k] for k in ('value','ms_per_step','kernel_ms','replay_windows_per_s')}); print(d['roofline']); print(d.get('deployed_budget')); print(d.get('counters'))"
