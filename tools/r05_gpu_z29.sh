#!/bin/bash
O=gpurun_out/r05z29; mkdir -p $O
python -m pytest tests/test_gpu_fuzz.py -m gpu -q --durations=3 > $O/fuzz_tests.log 2>&1; echo "rc $?" >> $O/fuzz_tests.log; tail -12 $O/fuzz_tests.log
