#!/bin/bash
O=gpurun_out/r05z21; mkdir -p $O
timeout 1700 python tests/dev/fuzz_td.py 120 0 > $O/fuzz_td.txt 2>&1; echo "rc $?" >> $O/fuzz_td.txt
grep "^tally\|^flagged" $O/fuzz_td.txt; sed -n '/^flagged/,$p' $O/fuzz_td.txt | head -40
