#!/bin/bash
O=gpurun_out/r05z40; mkdir -p $O
python tests/dev/replay_memory_watch.py 128 4 280 > $O/mem128.txt 2>&1; echo "rc $?" >> $O/mem128.txt; cat $O/mem128.txt | tail -12
python tests/dev/replay_memory_watch.py 8 2 330 > $O/mem8.txt 2>&1; echo "rc $?" >> $O/mem8.txt; cat $O/mem8.txt | tail -12
