#!/bin/bash
O=gpurun_out/r05z26; mkdir -p $O
timeout 900 python tests/dev/fuzz_lines.py 120 0 > $O/fuzz_lines.txt 2>&1; echo "rc $?" >> $O/fuzz_lines.txt
grep "^total\|^flagged" $O/fuzz_lines.txt; grep -A3 "matches differing [1-9]" $O/fuzz_lines.txt | head -30
python -m pytest tests/test_gpu_lines.py tests/test_gpu_replay.py -m gpu -x -q 2>&1 | tail -3
