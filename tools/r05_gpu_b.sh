#!/bin/bash
# round 5: the driver's suite on the new packer, FP64 micro-benchmark, lock-step replay laps at 8 / 32 / 128 streams
O=gpurun_out/r05b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python - > $O/fp64.txt 2>&1 <<'PY'
import sys, ctypes as C
sys.path.insert(0, "tc-viml_amd")
import tcv
L = tcv.lib()
for rep in range(3):
    o = (C.c_double * 4)()
    print("rc", L.tcv_microbench_fp64(o), "fma TF", o[0], "mfma TF", o[1], "CUs", o[2], "MHz", o[3])
PY
for S in 8 32 128; do
  TCV_DEBUG_PACK=1 python bench.py --mode replay --steps 40 --warmup 10 --streams $S > $O/replay_$S.json 2> $O/replay_$S.err
  grep "batch_create" $O/replay_$S.err | tail -4 > $O/replay_${S}_laps.txt
  rm -f $O/replay_$S.err
done
tail -3 $O/pytest.log; cat $O/fp64.txt
