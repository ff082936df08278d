#!/bin/bash
# What the register spills of tcv::marg_kernel<256> cost (round-4 review, item 3): A/B at EQUAL occupancy -- one workgroup per CU (TCV_MARG_GRID=256) --
# between the production build (two wavefronts per SIMD: 256 VGPRs, ~320 spilled) and the same source compiled for one wavefront per SIMD
# (build.py --margocc1: no VGPR spill).  Run through gpurun from the repo root.
for rep in 1 2; do
  for spec in "production (2 waves / SIMD budget):" "one wave / SIMD budget, no VGPR spill:TCV_LIB=tc-viml_amd/libtcv_hip_margocc1.so"; do
    name="${spec%%:*}"; var="${spec#*:}"
    env TCV_MARG_GRID=256 TCV_MARG_NT=256 $var python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-45s marginalisation %.3f ms (one workgroup per CU)' % ('$name', d['kernel_ms']['marginalize']))"
  done
done
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('production launch shape (two workgroups per CU)    marginalisation %.3f ms' % d['kernel_ms']['marginalize'])"
