#!/bin/bash
O=gpurun_out/r05i; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for S in 8 128; do for rep in 1 2 3; do
  python bench.py --mode replay --steps 60 --warmup 10 --streams $S --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['native_profile_ms_per_call']; print('streams $S: %6.0f windows/s  %s' % (d['value'], p))"
done; done > $O/replay.txt 2>&1
( time python bench.py ) > $O/bench.json 2> $O/bench.err
( time python bench.py --mode replay --steps 100 --warmup 10 ) > $O/bench_replay.json 2> $O/bench_replay.err
tail -3 $O/pytest.log; cat $O/replay.txt; tail -4 $O/bench.err; tail -4 $O/bench_replay.err
