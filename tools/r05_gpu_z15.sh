#!/bin/bash
O=gpurun_out/r05z15; mkdir -p $O
timeout 1200 python tools/replay_soak_euroc.py 8 300 100 > $O/soak.txt 2>&1; echo "rc $?" >> $O/soak.txt; tail -5 $O/soak.txt
timeout 600 python tools/replay_soak_euroc.py 4 300 60 > $O/soak60.txt 2>&1; echo "rc $?" >> $O/soak60.txt; tail -5 $O/soak60.txt
