#!/bin/bash
O=gpurun_out/r05z23; mkdir -p $O
timeout 1500 python tests/dev/fuzz_solve.py 600 1000 > $O/fuzz_solve_1000.txt 2>&1; echo "rc $?" >> $O/fuzz_solve_1000.txt
grep "^tally\|^mismatches" $O/fuzz_solve_1000.txt; sed -n '/^mismatches/,$p' $O/fuzz_solve_1000.txt | head -40
timeout 900 python tests/dev/fuzz_marg.py 400 1000 > $O/fuzz_marg_1000.txt 2>&1; echo "rc $?" >> $O/fuzz_marg_1000.txt
grep "^tally\|^flagged" $O/fuzz_marg_1000.txt; sed -n '/^flagged/,$p' $O/fuzz_marg_1000.txt | head -20
