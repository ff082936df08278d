#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/c; mkdir -p $O; cd $R
timeout 300 python3 tools/dev_small_batch.py 60 8 > $O/small_batch.txt 2>&1; cat $O/small_batch.txt
timeout 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -15 $O/pytest.log
