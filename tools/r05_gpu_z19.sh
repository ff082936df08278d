#!/bin/bash
O=gpurun_out/r05z19; mkdir -p $O
timeout 1200 python tests/dev/fuzz_marg.py 250 0 > $O/fuzz_marg.txt 2>&1; echo "rc $?" >> $O/fuzz_marg.txt
grep "^tally\|^flagged" $O/fuzz_marg.txt; sed -n '/^flagged/,$p' $O/fuzz_marg.txt | head -40
python -m pytest tests/test_gpu_solve.py -m gpu -q -k "rank_deficient" --tb=short 2>&1 | tail -30
