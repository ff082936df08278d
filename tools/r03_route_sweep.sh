#!/bin/bash
# Round 3, review item 1: the Amm+ route of the marginalisation decided by parity evidence.
#   bash tools/r03_route_sweep.sh          (on the GPU box through gpurun, from the repo root)
# 32 simulated association seeds and the full-length EuRoC table (three line modes x five sequences), once per route
# (unset: Cholesky when the rank is proven, =1: the reference's eigen pseudo-inverse everywhere).
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/route
mkdir -p $O
cd $R
SEEDS=$(seq 1 32)
python3 tests/dev/replay_seed_sweep.py $SEEDS > $O/seed_sweep_cholesky.txt 2>&1
TCV_MARG_EIG_MM=1 python3 tests/dev/replay_seed_sweep.py $SEEDS > $O/seed_sweep_eigen.txt 2>&1
python3 tests/dev/replay_euroc_full.py > $O/euroc_full_cholesky.log 2>&1; cp gpurun_out/euroc_full.json $O/euroc_full_cholesky.json
TCV_MARG_EIG_MM=1 python3 tests/dev/replay_euroc_full.py > $O/euroc_full_eigen.log 2>&1; cp gpurun_out/euroc_full.json $O/euroc_full_eigen.json
ls -la $O
