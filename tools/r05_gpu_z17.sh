#!/bin/bash
O=gpurun_out/r05z17; mkdir -p $O
{ for s in 258 167; do python tests/dev/fuzz_one.py $s 2>&1 | grep -v "worst entries\|point factor frames\|imu pairs"; done; } > $O/one.txt 2>&1
cat $O/one.txt
