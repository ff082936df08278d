#!/bin/bash
O=gpurun_out/r05z37; mkdir -p $O
one() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-46s solve %.4f ms  marginalisation %.4f ms  %.1f K solves/s' % ('$1', d['kernel_ms']['solve'], d['kernel_ms']['marginalize'], d['value'] / 1e3))"; }
for rep in 1 2 3 4; do
  one "panel with its refinement step (product)"
  TCV_LIB=tc-viml_amd/libtcv_hip_norefine.so one "build.py --norefine (explicit inverse alone)"
done > $O/ab.txt 2>&1
cat $O/ab.txt
