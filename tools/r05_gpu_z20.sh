#!/bin/bash
O=gpurun_out/r05z20; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
for r in 1 2 3; do python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', round(d['value']), d['kernel_ms'], 'single', d['single_window_ms']['solve_kernel'], 'replay', round(d['replay_windows_per_s']))"; done
