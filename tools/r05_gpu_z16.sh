#!/bin/bash
O=gpurun_out/r05z16; mkdir -p $O
timeout 1500 python tests/dev/fuzz_solve.py 400 0 > $O/fuzz.txt 2>&1; echo "rc $?" >> $O/fuzz.txt
tail -30 $O/fuzz.txt
python -m pytest tests/test_gpu_solve.py -m gpu -x -q -k "rank_deficient" 2>&1 | tail -5
