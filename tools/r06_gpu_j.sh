#!/bin/bash
# round 6, GPU call j: the seeded fuzzers of every row on the FINAL build (new seed ranges: 6000 ...): pre-integration (kernel re-cut this round),
# solve in four device configurations, marginalisation, Td, association, gauge fix, factor evaluators, the native estimator
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06j; mkdir -p $O; cd $R
run() { name=$1; shift; echo "tests/dev/$name.py $* on one MI355X (tools/r06_gpu_j.sh), final build of round 6" > $O/fuzz_$name.txt; timeout 1500 python3 tests/dev/$name.py "$@" >> $O/fuzz_$name.txt 2>&1; echo "rc $?" >> $O/fuzz_$name.txt; }
run fuzz_preint 200 6000
run fuzz_gauge 2000 6000
run fuzz_factors 2000 6000
run fuzz_lines 80 6000
run fuzz_marg 200 6000
run fuzz_td 60 6000
run fuzz_solve 300 6000
run fuzz_estimator 12 6000 60
for f in $O/fuzz_*.txt; do echo "== $f"; head -1 $f | cut -c1-160; grep -E "^worst|^flagged|^tally|^mismatches|^rc |identical|differ" $f | tail -6 | cut -c1-400; done
