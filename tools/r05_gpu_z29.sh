#!/bin/bash
O=gpurun_out/r05z29; mkdir -p $O
python -m pytest tests/test_gpu_fuzz.py -m gpu -q --durations=8 > $O/fuzz_tests.log 2>&1; echo "rc $?" >> $O/fuzz_tests.log; tail -16 $O/fuzz_tests.log
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
