#!/bin/bash
O=gpurun_out/r05z27; mkdir -p $O
timeout 600 python tests/dev/fuzz_gauge.py 2000 0 > $O/fuzz_gauge.txt 2>&1; echo "rc $?" >> $O/fuzz_gauge.txt
tail -30 $O/fuzz_gauge.txt
