#!/bin/bash
O=gpurun_out/r05z24; mkdir -p $O
timeout 1500 python tests/dev/fuzz_solve.py 300 2000 > $O/fuzz_solve_2000.txt 2>&1; echo "rc $?" >> $O/fuzz_solve_2000.txt
grep "^tally\|^mismatches" $O/fuzz_solve_2000.txt; sed -n '/^mismatches/,$p' $O/fuzz_solve_2000.txt | head -40; grep "^mixed" $O/fuzz_solve_2000.txt | head -5
