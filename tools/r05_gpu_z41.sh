#!/bin/bash
O=gpurun_out/r05z41; mkdir -p $O
timeout 1500 python tests/dev/fuzz_solve.py 400 100000 > $O/fuzz_solve_chained.txt 2>&1; echo "rc $?" >> $O/fuzz_solve_chained.txt
grep "^tally\|^mismatches" $O/fuzz_solve_chained.txt; sed -n '/^mismatches/,$p' $O/fuzz_solve_chained.txt | head -40; grep -c "chained" $O/fuzz_solve_chained.txt
timeout 900 python tests/dev/fuzz_marg.py 300 100000 > $O/fuzz_marg_chained.txt 2>&1; echo "rc $?" >> $O/fuzz_marg_chained.txt
grep "^tally\|^flagged" $O/fuzz_marg_chained.txt; sed -n '/^flagged/,$p' $O/fuzz_marg_chained.txt | head -20
