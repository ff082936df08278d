#!/bin/bash
# round 6, GPU call n: closing checks at the round's last commit -- the whole GPU suite; the estimator's separate-gauge path (TCV_EST_SEPARATE_GAUGE=1,
# the A/B partner of the fused epilogue) under the replay and gauge tests; the marginalisation and estimator fuzzers on seed ranges not run before
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06n; mkdir -p $O; cd $R
(python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > $O/tests_all.txt
(TCV_EST_SEPARATE_GAUGE=1 python -m pytest tests/test_gpu_replay.py tests/test_gpu_gauge.py tests/test_gpu_teacher.py -x -q 2>&1 | tail -4) > $O/tests_separate_gauge.txt
run() { name=$1; shift; echo "tests/dev/$name.py $* on one MI355X (tools/r06_gpu_n.sh), last commit of round 6" > $O/fuzz_$name.txt; timeout 1500 python3 tests/dev/$name.py "$@" >> $O/fuzz_$name.txt 2>&1; tail -2 $O/fuzz_$name.txt; }
run fuzz_marg 400 6200
run fuzz_estimator 8 6100 60
run fuzz_solve 200 6400
cat $O/tests_all.txt $O/tests_separate_gauge.txt
