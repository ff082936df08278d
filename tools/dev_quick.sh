#!/bin/bash
# quick GPU regression: parity checks + phase profile + bench line (developer loop)
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -x -q -m gpu > gpurun_out/q_pytest.log 2>&1; echo "pytest exit=$?" >> gpurun_out/q_pytest.log
TCV_LIB=tc-viml_amd/libtcv_hip_prof.so timeout 200 python tools/dev_phase_profile.py 256 256 --prior > gpurun_out/q_prof.log 2>&1
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/q_bench.log 2>&1
