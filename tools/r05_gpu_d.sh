#!/bin/bash
# round 5: the new bench lines (default with roofline_fp64 / batch_sweep / run_to_convergence / replay CPU baseline; replay with cpu_baseline + roofline), replay laps
O=gpurun_out/r05d; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
( time python bench.py ) > $O/bench.json 2> $O/bench.err
( time python bench.py --mode replay --steps 100 --warmup 10 ) > $O/bench_replay.json 2> $O/bench_replay.err
for S in 8 128; do
  python bench.py --mode replay --steps 40 --warmup 10 --streams $S --no-cpu-baseline > $O/replay_$S.json 2> /dev/null
done
tail -3 $O/pytest.log; tail -4 $O/bench.err; tail -4 $O/bench_replay.err
