#!/bin/bash
# round 3, GPU call A: the GPU suite (with the new line-factor replays), small-batch latency + phase tables, baseline bench
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/a; mkdir -p $O; cd $R
python3 -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python3 tools/dev_small_batch.py 60 12 > $O/small_batch.txt 2>&1
TCV_LIB=tc-viml_amd/libtcv_hip_prof.so python3 tools/dev_small_batch.py 60 4 > $O/small_batch_prof.txt 2>&1
python3 bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
tail -3 $O/pytest.log; cat $O/small_batch.txt; cat $O/bench.json
