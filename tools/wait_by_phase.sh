#!/bin/bash
# Round 6, review item 1(a): the wait of the solve kernel PER PHASE -- hardware counters with one phase removed at a time (ablation build, run-time mask:
# the registers of the removed phase stay, so residency and spills are the production kernel's): wavefront cycles, the wait / issue split, vector-memory
# instructions and their mean latency; phase = full - without.  Run on the GPU box from the repo root:  bash tools/wait_by_phase.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export TCV_LIB=$R/tc-viml_amd/libtcv_hip_abl.so
O=$R/gpurun_out/wait_by_phase
mkdir -p $O
for NM in full:0 vis_eval:0x1 vis_gather:0x2 schur:0x8 prior:0x110 imu_raw:0x20 imu_whiten:0x40 imu_gather:0x80 fin:0x600 chain_fwd:0x800 chol:0x1000 back:0x2000 chain_bwd:0x4000 linearise_all:0x1ff solve_all:0xfe00 everything:0x7ffff; do
  N=${NM%%:*}; M=${NM##*:}
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES -d $O/$N.a -o p --output-format csv -- python3 $R/tools/dev_ablate_one.py $M 1024 3 > $O/$N.a.log 2>&1
  rocprofv3 --kernel-trace --pmc VmemLatency -d $O/$N.b -o p --output-format csv -- python3 $R/tools/dev_ablate_one.py $M 1024 3 > $O/$N.b.log 2>&1
  rocprofv3 --kernel-trace --pmc LdsLatency -d $O/$N.c -o p --output-format csv -- python3 $R/tools/dev_ablate_one.py $M 1024 3 > $O/$N.c.log 2>&1
  tail -1 $O/$N.a.log
done
python3 - <<PY > $R/gpurun_out/wait_by_phase.txt 2>&1
import csv, glob, os, collections
O = "$O"
res = collections.OrderedDict(); ms = {}
order = ["full", "vis_eval", "vis_gather", "schur", "prior", "imu_raw", "imu_whiten", "imu_gather", "fin", "chain_fwd", "chol", "back", "chain_bwd", "linearise_all", "solve_all", "everything"]
for n in order:
    acc = collections.defaultdict(list)
    for sub in ("a", "b", "c"):
        for f in glob.glob(f"{O}/{n}.{sub}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "solve_kernel" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[n] = {k: sum(v) / len(v) for k, v in acc.items()}
    try:
        ms[n] = float(open(f"{O}/{n}.a.log").read().strip().splitlines()[-1].split()[-1])
    except Exception:
        ms[n] = float("nan")
f = res["full"]
print("solve kernel, B = 1024, every step accepted (ablation build, run-time mask); per launch; cycles in quad-cycles x 1e9, instructions x 1e6")
hdr = "%-14s %8s %9s %9s %9s %9s %9s %9s %9s %10s" % ("without", "ms", "wave cyc", "wait", "issue st", "issuing", "VMEM", "VMEM lat", "LDS lat", "VMEM in fl")
print(hdr)
def row(n, r, m):
    vm = r.get("SQ_INSTS_VMEM", 0); vl = r.get("VmemLatency", 0)
    return "%-14s %8.3f %9.3f %9.3f %9.3f %9.3f %9.2f %9.0f %9.0f %10.3f" % (n, m, r.get("SQ_WAVE_CYCLES", 0) / 1e9, r.get("SQ_WAIT_ANY", 0) / 1e9, r.get("SQ_WAIT_INST_ANY", 0) / 1e9,
                                                                    r.get("SQ_ACTIVE_INST_ANY", 0) / 1e9, vm / 1e6, vl, r.get("LdsLatency", 0), vm * vl / 4 / 1e9)
for n, r in res.items():
    print(row(n, r, ms[n]))
print()
print("phase = full - without  (VMEM in flight = instructions x mean latency, in the same quad-cycle unit as the wait: an upper bound of the wait the phase's memory accesses explain)")
print("%-14s %8s %9s %9s %9s %9s %9s %10s %12s" % ("phase", "ms", "wave cyc", "wait", "issue st", "issuing", "VMEM", "VMEM in fl", "wait share"))
for n, r in res.items():
    if n == "full":
        continue
    d = lambda k: f.get(k, 0) - r.get(k, 0)
    infl = (f.get("SQ_INSTS_VMEM", 0) * f.get("VmemLatency", 0) - r.get("SQ_INSTS_VMEM", 0) * r.get("VmemLatency", 0)) / 4
    print("%-14s %8.3f %9.3f %9.3f %9.3f %9.3f %9.2f %10.3f %11.1f %%" % (n, ms["full"] - ms[n], d("SQ_WAVE_CYCLES") / 1e9, d("SQ_WAIT_ANY") / 1e9, d("SQ_WAIT_INST_ANY") / 1e9, d("SQ_ACTIVE_INST_ANY") / 1e9,
                                                                      d("SQ_INSTS_VMEM") / 1e6, infl / 1e9, 100 * d("SQ_WAIT_ANY") / max(f.get("SQ_WAIT_ANY", 1), 1)))
PY
cat $R/gpurun_out/wait_by_phase.txt
rm -rf $O/*/
