#!/bin/bash
# round 5, first GPU call: the driver's suite on the advisor fixes, then where tcv_batch_create spends a lock-step frame (8 / 128 streams)
mkdir -p gpurun_out/r05a
python -m pytest tests -m gpu -x -q > gpurun_out/r05a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05a/pytest.log
for S in 8 128; do
  TCV_DEBUG_PACK=1 python bench.py --mode replay --steps 30 --warmup 10 --streams $S > gpurun_out/r05a/replay_$S.json 2> gpurun_out/r05a/replay_$S.err
  grep "batch_create" gpurun_out/r05a/replay_$S.err | tail -8 > gpurun_out/r05a/replay_${S}_laps.txt
  rm -f gpurun_out/r05a/replay_$S.err
done
tail -3 gpurun_out/r05a/pytest.log
