#!/bin/bash
O=gpurun_out/r05z12; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
for S in 8 128; do python bench.py --mode replay --steps 60 --warmup 8 --streams $S --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['streams'], d['config']['host_threads'], round(d['value']), d['host_cpu']['cores_busy_mean'])"; done
