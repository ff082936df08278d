#!/bin/bash
# (provenance of profiles/r06_preint_scan.txt: preint_scan_kernel and its TCV_PREINT_SEQ switch were removed again after this measurement)
# round 6, GPU call m: pre-integration as a scan over the samples (preint_scan_kernel) against the sample-by-sample kernel (TCV_PREINT_SEQ=1): parity, kernel
# durations in a replay, throughput
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06m; mkdir -p $O; cd $R
(python -m pytest tests/test_gpu_preint.py tests/test_gpu_fuzz.py tests/test_gpu_replay.py tests/test_gpu_teacher.py tests/test_gpu_resident.py -x -q 2>&1 | tail -6) > $O/tests.txt
(echo "tests/dev/fuzz_preint.py 200 7000, scan kernel"; python3 tests/dev/fuzz_preint.py 200 7000 2>&1 | grep -E "^worst|^flagged" ) > $O/fuzz_preint.txt
python3 tools/dev_preint_time.py > $O/preint_time_scan.txt 2>&1
TCV_PREINT_SEQ=1 python3 tools/dev_preint_time.py > $O/preint_time_seq.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for V in scan seq; do
  if [ $V = seq ]; then export TCV_PREINT_SEQ=1; else unset TCV_PREINT_SEQ; fi
  for S in 8 128; do
    T=2; [ $S = 128 ] && T=4
    rocprofv3 --kernel-trace --stats -d $O/rp_${V}_$S -o r --output-format csv -- python3 $R/bench.py --mode replay --streams $S --host-threads $T --steps 40 --warmup 8 > $O/rp_${V}_$S.log 2>&1
    f=$(find $O/rp_${V}_$S -name "*kernel_stats.csv" | head -1)
    echo "== $V kernel, $S streams: $(tail -1 $O/rp_${V}_$S.log | python3 -c "import json,sys; print('%.0f windows/s' % json.loads(sys.stdin.read())['value'])")" >> $O/replay_kernels.txt
    grep -E "preint|solve_kernel|marg_kernel|lines_match" $f | cut -d, -f1-4 >> $O/replay_kernels.txt
    rm -rf $O/rp_${V}_$S
  done
done
unset TCV_PREINT_SEQ
cat $O/tests.txt $O/fuzz_preint.txt; echo scan; cat $O/preint_time_scan.txt; echo seq; cat $O/preint_time_seq.txt; cat $O/replay_kernels.txt
