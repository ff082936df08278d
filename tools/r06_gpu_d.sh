#!/bin/bash
# (the register variants need the -DTCV_MARG_REG_TRIDIAG build since the end of the round: TCV_LIB=tc-viml_amd/libtcv_hip_regtri.so)
# round 6, GPU call d: four-wave register-resident tridiagonalisation (parity + A/B + phase cycles), the re-cut pre-integration kernel, the
# boundary tests of the round, the phase-split bound with the fixed tool
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06d; mkdir -p $O; cd $R
(python -m pytest tests/test_gpu_marg.py tests/test_gpu_bench_shape.py tests/test_gpu_preint.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -15) > $O/tests_a.txt
(python -m pytest tests/test_gpu_replay.py -x -q 2>&1 | tail -15) > $O/tests_replay.txt
for rep in 1 2; do
  for spec in "registers, four wavefronts:" "registers, two wavefronts:TCV_MARG_EIG_FLAGS=8" "LDS-resident (round 5):TCV_MARG_EIG_FLAGS=4"; do
    name="${spec%%:*}"; var="${spec#*:}"
    env $var python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-32s solve %.3f ms  marginalisation %.3f ms  %.1f K solves/s' % ('$name', d['kernel_ms']['solve'], d['kernel_ms']['marginalize'], d['value'] / 1e3))"
  done
done > $O/marg_ab.txt 2>&1
for F in 0 8 4; do
  echo "== TCV_MARG_EIG_FLAGS=$F, 256 threads, one workgroup per CU (8 windows)" >> $O/phase_cycles_marg.txt
  TCV_MARG_EIG_FLAGS=$F TCV_MARG_NT=256 TCV_LIB=tc-viml_amd/libtcv_hip_prof.so TCV_DEBUG=1 python3 tools/dev_marg_profile.py 2>&1 | grep -E "window 0|eig_rr|schur|proj  |imu  |out  " | head -12 >> $O/phase_cycles_marg.txt
  echo "== TCV_MARG_EIG_FLAGS=$F, 512 threads" >> $O/phase_cycles_marg.txt
  TCV_MARG_EIG_FLAGS=$F TCV_MARG_NT=512 TCV_LIB=tc-viml_amd/libtcv_hip_prof.so TCV_DEBUG=1 python3 tools/dev_marg_profile.py 2>&1 | grep -E "window 0|eig_rr" | head -9 >> $O/phase_cycles_marg.txt
done
python3 tools/dev_preint_time.py > $O/preint_time.txt 2>&1
python3 tools/dev_single_latency.py > $O/single_latency.txt 2>&1
python3 tools/dev_phase_split.py --frames 6 --windows 3072 --lds 6400 > $O/phase_split_lds6400.txt 2>&1
python3 bench.py --mode replay --steps 100 --warmup 10 > $O/bench_replay.json 2> /dev/null
python3 bench.py --mode replay --streams 128 --steps 40 --warmup 8 > $O/bench_replay128.json 2> /dev/null
cat $O/tests_a.txt $O/tests_replay.txt $O/marg_ab.txt $O/phase_cycles_marg.txt $O/preint_time.txt $O/phase_split_lds6400.txt; tail -12 $O/single_latency.txt
python3 -c "import json; [print(f, json.load(open('$O/'+f))['value']) for f in ('bench_replay.json','bench_replay128.json')]"
