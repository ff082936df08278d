#!/bin/bash
O=gpurun_out/r05z6; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
python bench.py > $O/bench_final.json 2> $O/bench_final.err; tail -c 600 $O/bench_final.json
python bench.py --mode replay --streams 8 --host-threads 2 --steps 100 --warmup 10 > $O/bench_replay_final.json 2> $O/bench_replay_final.err; tail -c 300 $O/bench_replay_final.json
python bench.py --mode replay --streams 128 --steps 60 --warmup 10 > $O/bench_replay128_final.json 2> $O/bench_replay128_final.err; tail -c 300 $O/bench_replay128_final.json
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log; tail -2 $O/smoke.log
