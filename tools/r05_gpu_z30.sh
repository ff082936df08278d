#!/bin/bash
O=gpurun_out/r05z30; mkdir -p $O
timeout 1500 python tests/dev/fuzz_estimator.py 24 0 60 > $O/fuzz_estimator.txt 2>&1; echo "rc $?" >> $O/fuzz_estimator.txt
tail -40 $O/fuzz_estimator.txt
