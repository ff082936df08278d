#!/bin/bash
O=gpurun_out/r05z46; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
python tests/dev/replay_memory_watch.py 32 2 60 2>&1 | tail -3
