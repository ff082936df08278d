#!/bin/bash
O=gpurun_out/r05z50; mkdir -p $O
timeout 500 python tests/dev/thread_churn.py 16 4 150 > $O/churn.txt 2>&1 < /dev/null; echo "rc $?" >> $O/churn.txt; tail -9 $O/churn.txt
