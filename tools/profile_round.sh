#!/bin/bash
# Collects the evidence files of profiles/ on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <tag>
# rocprofv3 --kernel-trace --stats of the benchmark command, separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ counters;
# kernel-trace only, as gpurun requires), the bench line itself and the per-phase cycle tables of the -DTCV_PROFILE build.
TAG=${1:-r06}
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
# round 6: where the wavefronts wait -- the split of SQ_WAVE_CYCLES, and mean latencies by Little's law (rocprofv3's accumulate() derived counters; one per pass)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SMEM -d $O/sq4 -o sq --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/sq4.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES -d $O/sq5 -o sq --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/sq5.log 2>&1
for LC in VmemLatency LdsLatency SmemLatency; do
  rocprofv3 --kernel-trace --pmc $LC -d $O/lat_$LC -o l --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/lat_$LC.log 2>&1
done
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/tcc -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/tcc.log 2>&1
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
out = {"sq": summarise("sq"), "sq2": summarise("sq2"), "mfma": summarise("sq3"), "wait": summarise("sq4"), "active": summarise("sq5"), "tcc": summarise("tcc"),
       "latency": {k: v for d in ("lat_VmemLatency", "lat_LdsLatency", "lat_SmemLatency") for k, v in summarise(d).items()}}
def merged(*ds):
    m = {}
    for d in ds:
        for k, v in d.items():
            m.setdefault(k, {}).update(v)
    return m
out["latency"] = merged(summarise("lat_VmemLatency"), summarise("lat_LdsLatency"), summarise("lat_SmemLatency"))
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
wt, ac, lt, tc = pick(out["wait"], sk), pick(out["active"], sk), pick(out["latency"], sk), pick(out["tcc"], sk)
import hashlib, glob as _g
def csrc_sha16(root):      # what bench.py compares with: the counters belong to THESE kernel sources
    h = hashlib.sha256()
    for f in sorted(_g.glob(os.path.join(root, "tc-viml_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
def frac(x, k, d):
    return (x.get(k, 0) / x[d]) if x.get(d) else None
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
       "solve_kernel_write_bytes_per_launch": trk.get("write_bytes_per_launch_raw"), "solve_kernel_fetch_bytes_per_launch_raw": trk.get("fetch_bytes_per_launch_raw"),
       "csrc_sha16": csrc_sha16("$R"),
       # executed FP64 work of one launch: matrix-core ops x 512 flops (SQ_INSTS_VALU_MFMA_MOPS_F64), and an UPPER bound of the vector part
       # (every VALU instruction priced as a 64-lane FMA)
       "mfma_mops_f64_per_launch": c.get("SQ_INSTS_VALU_MFMA_MOPS_F64"), "valu_insts_per_launch": a.get("SQ_INSTS_VALU"),
       # where the wavefronts' cycles go (disjoint: waiting at s_waitcnt / barrier | issue stalls | issuing) and what they issue
       "wait_split": {"wait_any": frac(wt, "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), "wait_inst_any": frac(wt, "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
                      "wait_inst_lds": frac(wt, "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"), "active_inst_any": frac(wt, "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"),
                      "active_valu": frac(ac, "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"), "active_scalar": frac(ac, "SQ_ACTIVE_INST_SCA", "SQ_WAVE_CYCLES"),
                      "active_lds": frac(ac, "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES"), "active_flat": frac(ac, "SQ_ACTIVE_INST_FLAT", "SQ_WAVE_CYCLES")},
       "mean_latency_cycles": {"vmem": lt.get("VmemLatency"), "lds": lt.get("LdsLatency"), "smem": lt.get("SmemLatency")},
       "insts_per_launch": {"vmem": wt.get("SQ_INSTS_VMEM"), "lds": b2.get("SQ_INSTS_LDS"), "smem": wt.get("SQ_INSTS_SMEM"), "waves": wt.get("SQ_WAVES")},
       "l2_hit_rate": (tc.get("TCC_HIT_sum", 0) / (tc.get("TCC_HIT_sum", 0) + tc.get("TCC_MISS_sum", 0))) if (tc.get("TCC_HIT_sum", 0) + tc.get("TCC_MISS_sum", 0)) else None}
json.dump(cnt, open(f"{O}/counters.json", "w"), indent=1)
PY
ls $O
# round 4: kernel durations of a lock-step replay frame, thread sweep of the replay
bash $R/tools/dev_replay_kernel_trace.sh > /dev/null 2>&1; cp $R/gpurun_out/replay_kernel_stats.csv $O/replay_kernel_stats.csv 2>/dev/null
THREADS="1 2 4" DEVSTATE="1" bash $R/tools/dev_replay_thread_sweep.sh > $O/replay_thread_sweep.txt 2>&1
GPU_MAX_HW_QUEUES=8 QUEUES=8 THREADS="4" DEVSTATE="1" bash $R/tools/dev_replay_thread_sweep.sh >> $O/replay_thread_sweep.txt 2>&1
