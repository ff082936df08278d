#!/bin/bash
O=gpurun_out/r05z18; mkdir -p $O
{ for s in 156 240 58 4; do python tests/dev/fuzz_one.py $s 2>&1 | grep -v "worst entries\|costs gpu\|mu / radius\|point factor frames\|imu pairs"; done; } > $O/one.txt 2>&1
cat $O/one.txt
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', round(d['value']), d['kernel_ms'])"
