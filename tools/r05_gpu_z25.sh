#!/bin/bash
O=gpurun_out/r05z25; mkdir -p $O
timeout 900 python tests/dev/fuzz_lines.py 120 0 > $O/fuzz_lines.txt 2>&1; echo "rc $?" >> $O/fuzz_lines.txt
grep "^total\|^flagged" $O/fuzz_lines.txt; grep -A3 "matches differing [1-9]" $O/fuzz_lines.txt | head -30
timeout 900 python tests/dev/fuzz_preint.py 60 0 > $O/fuzz_preint.txt 2>&1; echo "rc $?" >> $O/fuzz_preint.txt
tail -12 $O/fuzz_preint.txt
