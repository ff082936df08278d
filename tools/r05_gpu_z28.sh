#!/bin/bash
O=gpurun_out/r05z28; mkdir -p $O
timeout 900 python tests/dev/fuzz_factors.py 3000 0 > $O/fuzz_factors.txt 2>&1; echo "rc $?" >> $O/fuzz_factors.txt
tail -30 $O/fuzz_factors.txt
