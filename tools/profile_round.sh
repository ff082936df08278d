#!/bin/bash
# Collects the evidence files of profiles/ on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <tag>
# rocprofv3 --kernel-trace --stats of the benchmark command, separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ counters;
# kernel-trace only, as gpurun requires), the bench line itself and the per-phase cycle tables of the -DTCV_PROFILE build.
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/sq -o sq --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA -d $O/sq2 -o sq --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/sq2.log 2>&1
# north_star's "MFMA-busy": cycles the matrix cores are busy against the cycles the CUs are busy (own pass; SQ_VALU_MFMA_BUSY_CYCLES counts cycles)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 -d $O/sq3 -o sq --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/sq3.log 2>&1
cd $R
python3 tools/pmc_traffic.py $O/fetch $O/write $O/pmc_traffic.json > /dev/null 2>&1
if [ -f tc-viml_amd/libtcv_hip_prof.so ]; then
  TCV_LIB=tc-viml_amd/libtcv_hip_prof.so python3 tools/dev_phase_profile.py 512 256 --prior > $O/phase_cycles_solve.txt 2>&1
  TCV_LIB=tc-viml_amd/libtcv_hip_prof.so TCV_DEBUG=1 python3 tools/dev_marg_profile.py > $O/phase_cycles_marg.txt 2>&1
  TCV_MARG_NT=256 TCV_LIB=tc-viml_amd/libtcv_hip_prof.so TCV_DEBUG=1 python3 tools/dev_marg_profile.py > $O/phase_cycles_marg_256.txt 2>&1
fi
if [ -f tc-viml_amd/libtcv_hip_abl.so ]; then
  TCV_LIB=tc-viml_amd/libtcv_hip_abl.so python3 tools/dev_ablate.py 1024 > $O/ablation_solve_B1024.txt 2>&1
fi
python3 tests/dev/stream_breakdown.py > $O/stream_breakdown.txt 2>&1
python3 tests/dev/parity_sweep.py 1024 > $O/parity_sweep.txt 2>&1
python3 tests/dev/marg_floor.py > $O/marg_floor.txt 2>&1
# round 3: small-batch latency (cooperative mode) with the per-phase table of the -DTCV_PROFILE build, replay and stream modes, replay frame accounting
python3 tools/dev_small_batch.py 60 12 > $O/small_batch.txt 2>&1
if [ -f tc-viml_amd/libtcv_hip_prof.so ]; then
  TCV_DEBUG=1 TCV_LIB=tc-viml_amd/libtcv_hip_prof.so python3 tools/dev_small_batch.py 60 3 > $O/small_batch_prof.txt 2>&1
fi
python3 bench.py --mode replay --steps 150 --warmup 10 > $O/bench_replay.json 2> /dev/null
python3 bench.py --mode replay --steps 150 --warmup 10 --host-threads 1 > $O/bench_replay_1thread.json 2> /dev/null
python3 bench.py --mode stream --windows 2048 > $O/bench_stream_2048.json 2> /dev/null
python3 tools/replay_euroc.py --native --profile --frames 340 --out $O/euroc > $O/replay_euroc_native.json 2> /dev/null
# round 3: A/B of the round's kernel changes on the final build (developer switches), the experiments of DESIGN 6b
for spec in "default:" "round-2 eigenvalue search:TCV_MARG_EIG_FLAGS=1" "reflector-by-reflector back-transformation:TCV_MARG_EIG_FLAGS=2" "both round-2 eigen paths:TCV_MARG_EIG_FLAGS=3" "prior with its zero rows:TCV_PRIOR_FULL=1"; do
  name="${spec%%:*}"; var="${spec#*:}"
  for rep in 1 2; do
    env $var python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-45s solve %.3f ms  marginalisation %.3f ms  %.1f K solves/s' % ('$name', d['kernel_ms']['solve'], d['kernel_ms']['marginalize'], d['value'] / 1e3))"
  done
done > $O/kernel_ab.txt 2>&1
find $O -name "*_kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
python3 - <<PY
import csv, glob, collections, json
O = "$O"
def summarise(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items() if "tcv::" in k}
out = {"sq": summarise("sq"), "sq2": summarise("sq2"), "mfma": summarise("sq3")}
json.dump(out, open(f"{O}/sq_counters.json", "w"), indent=1)
# the figures bench.py copies onto its default line (profiles/counters.json), with the commit they were taken at
import os
def pick(d, pat):
    for k, v in d.items():
        if pat in k:
            return v
    return {}
sk = "solve_kernel<256, true, true, false, false>"
a, b2, c = pick(out["sq"], sk), pick(out["sq2"], sk), pick(out["mfma"], sk)
tr = {}
try:
    tr = json.load(open(f"{O}/pmc_traffic.json"))
except (OSError, ValueError):
    pass
trk = pick(tr.get("kernels", {}), sk)
cnt = {"commit": os.environ.get("COMMIT", "unknown"), "source": "tools/profile_round.sh: rocprofv3 --kernel-trace --pmc passes of python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras (B = 1024)",
       "kernel": sk,
       "valu_active": (a.get("SQ_ACTIVE_INST_VALU", 0) / a["SQ_WAVE_CYCLES"]) if a.get("SQ_WAVE_CYCLES") else None,
       "waiting": (a.get("SQ_WAIT_ANY", 0) / a["SQ_WAVE_CYCLES"]) if a.get("SQ_WAVE_CYCLES") else None,
       "mfma_busy": (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / c["SQ_BUSY_CU_CYCLES"]) if c.get("SQ_BUSY_CU_CYCLES") else None,
       "salu_per_valu": (b2.get("SQ_INSTS_SALU", 0) / a["SQ_INSTS_VALU"]) if a.get("SQ_INSTS_VALU") else None,
       "solve_kernel_hbm_bytes_per_launch": tr.get("solve_kernel_hbm_bytes_per_launch"), "marg_kernel_hbm_bytes_per_launch": tr.get("marg_kernel_hbm_bytes_per_launch"),
       "solve_kernel_write_bytes_per_launch": trk.get("write_bytes_per_launch_raw"), "solve_kernel_fetch_bytes_per_launch_raw": trk.get("fetch_bytes_per_launch_raw")}
json.dump(cnt, open(f"{O}/counters.json", "w"), indent=1)
PY
ls $O
# round 4: kernel durations of a lock-step replay frame, thread sweep of the replay
bash $R/tools/dev_replay_kernel_trace.sh > /dev/null 2>&1; cp $R/gpurun_out/replay_kernel_stats.csv $O/replay_kernel_stats.csv 2>/dev/null
THREADS="1 2 4" DEVSTATE="1" bash $R/tools/dev_replay_thread_sweep.sh > $O/replay_thread_sweep.txt 2>&1
GPU_MAX_HW_QUEUES=8 QUEUES=8 THREADS="4" DEVSTATE="1" bash $R/tools/dev_replay_thread_sweep.sh >> $O/replay_thread_sweep.txt 2>&1
