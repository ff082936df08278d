#!/bin/bash
O=gpurun_out/r05g; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python -m pytest tests/test_gpu_teacher.py -x -q -s -k "native" > $O/pytest_native.log 2>&1
for S in 8 32 128; do for D in 0 1 12; do
  TCV_EST_MARG_DEFER=$D python bench.py --mode replay --steps 60 --warmup 10 --streams $S --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; print('streams $S defer-from $D: %6.0f windows/s  kernels %.3f batch_create %.3f assoc+ %.3f downloads %.3f' % (d['value'], p['kernels'], p['batch_create'], p['assoc+triangulate+window'], p['downloads']))"
done; done > $O/defer_threshold.txt 2>&1
bash tools/r05_writes_by_phase.sh > $O/writes_by_phase.log 2>&1
cp gpurun_out/abl_wr_table.txt $O/
tail -4 $O/pytest.log; grep -E "native windows|passed|failed|Error" $O/pytest_native.log | tail -6; cat $O/defer_threshold.txt; cat $O/abl_wr_table.txt
