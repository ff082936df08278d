#!/bin/bash
# round 6, GPU call c: (1) the column-owner tridiagonalisation of marg_kernel: parity suites, A/B against the LDS-resident path (TCV_MARG_EIG_FLAGS=4),
# phase cycles of the profile build; (2) the occupancy-3 experiment again with LDS sizes that leave room for the allocation granularity, with the
# residency actually reached (MeanOccupancyPerActiveCU) measured beside every timing
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c; mkdir -p $O; cd $R
(python -m pytest tests/test_gpu_marg.py tests/test_gpu_bench_shape.py tests/test_gpu_resident.py -x -q 2>&1 | tail -15) > $O/marg_tests.txt
for rep in 1 2; do
  for spec in "columns-in-registers:" "LDS-resident (round 5):TCV_MARG_EIG_FLAGS=4"; do
    name="${spec%%:*}"; var="${spec#*:}"
    env $var python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-32s solve %.3f ms  marginalisation %.3f ms  %.1f K solves/s' % ('$name', d['kernel_ms']['solve'], d['kernel_ms']['marginalize'], d['value'] / 1e3))"
  done
done > $O/marg_ab.txt 2>&1
TCV_MARG_NT=256 TCV_LIB=tc-viml_amd/libtcv_hip_prof.so TCV_DEBUG=1 python3 tools/dev_marg_profile.py > $O/phase_cycles_marg_256.txt 2>&1
for L in 6784 6528 6400; do
  OCC3_LDS_DOUBLES=$L python3 tools/dev_occupancy3.py --frames 6 --windows 1536 > $O/occ3_lds$L.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
for L in 6784 6528 6400; do
  export TCV_CHAIN_LDS_DOUBLES=$L TCV_LIB=$R/tc-viml_amd/libtcv_hip_occ3.so TCV_GRID=768
  rocprofv3 --kernel-trace --pmc MeanOccupancyPerActiveCU -d $O/occ_$L -o p --output-format csv -- python3 $R/tools/dev_occupancy3.py --worker 6 1536 > $O/occ_$L.log 2>&1
  python3 - <<PY >> $O/occ3_residency.txt
import csv, glob
v = [float(r["Counter_Value"]) for f in glob.glob("$O/occ_$L/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "solve_kernel" in r["Kernel_Name"]]
print("TCV_CHAIN_LDS_DOUBLES=$L grid 768 (3 per CU asked): MeanOccupancyPerActiveCU %.3f over %d launches" % (sum(v) / max(1, len(v)), len(v)))
PY
  rm -rf $O/occ_$L
done
unset TCV_CHAIN_LDS_DOUBLES TCV_LIB TCV_GRID
cd $R
for L in 6400 4736; do python3 tools/dev_phase_split.py --frames 6 --windows 3072 --lds $L > $O/phase_split_lds$L.txt 2>&1; done
python3 tools/dev_phase_split.py --frames 4 --windows 3072 --lds 4736 > $O/phase_split_4frames_lds4736.txt 2>&1
cd /tmp
for cfg in lin3:768:6400 sol3:768:6400 lin4:1024:4736 sol4:1024:4736; do
  IFS=: read lib grid L <<< "$cfg"
  export TCV_CHAIN_LDS_DOUBLES=$L TCV_LIB=$R/tc-viml_amd/libtcv_hip_$lib.so TCV_GRID=$grid
  rocprofv3 --kernel-trace --pmc MeanOccupancyPerActiveCU -d $O/occ_$lib -o p --output-format csv -- python3 $R/tools/dev_occupancy3.py --worker 4 3072 > $O/occ_$lib.log 2>&1
  python3 - <<PY >> $O/phase_split_residency.txt
import csv, glob
v = [float(r["Counter_Value"]) for f in glob.glob("$O/occ_$lib/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "solve_kernel" in r["Kernel_Name"]]
print("$lib grid $grid lds $L doubles (4-frame windows): MeanOccupancyPerActiveCU %.3f over %d launches" % (sum(v) / max(1, len(v)), len(v)))
PY
  rm -rf $O/occ_$lib
done
unset TCV_CHAIN_LDS_DOUBLES TCV_LIB TCV_GRID
cat $O/phase_split*.txt
cat $O/marg_tests.txt $O/marg_ab.txt $O/occ3_residency.txt; tail -8 $O/occ3_lds*.txt; grep -E "eig_rr|tridiag" $O/phase_cycles_marg_256.txt | head -8
