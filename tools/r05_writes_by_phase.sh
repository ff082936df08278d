# round 5 (review item 4 i): which phase of the solve kernel writes the 1.6 GB per launch -- HBM write / fetch bytes and VMEM write instructions of the
# kernel with one phase removed at a time (ablation build; separate --pmc passes for the TCC counters, kernel-trace only).  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export TCV_LIB=$R/tc-viml_amd/libtcv_hip_abl.so
O=$R/gpurun_out/abl_wr
mkdir -p $O
for NM in full:0 vis_eval:0x1 vis_gather:0x2 schur:0x8 prior_A:0x10 imu_raw:0x20 imu_gather:0x80 fin_scale:0x200 chain_fwd:0x800 chol:0x1000 back:0x2000 everything:0x7ffff; do
  N=${NM%%:*}; M=${NM##*:}
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/$N.w -o p --output-format csv -- python3 $R/tools/dev_ablate_one.py $M 1024 3 > $O/$N.w.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/$N.f -o p --output-format csv -- python3 $R/tools/dev_ablate_one.py $M 1024 3 > $O/$N.f.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM -d $O/$N.i -o p --output-format csv -- python3 $R/tools/dev_ablate_one.py $M 1024 3 > $O/$N.i.log 2>&1
  tail -1 $O/$N.i.log
done
python3 - $O > $R/gpurun_out/abl_wr_table.txt 2>&1 <<'PY'
import csv, glob, os, sys
root = sys.argv[1]
CNT = ["WRITE_SIZE", "FETCH_SIZE", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM", "SQ_INSTS_SALU", "SQ_INSTS_VALU"]
res = {}
for d in sorted(os.listdir(root)):
    p = os.path.join(root, d)
    if not os.path.isdir(p):
        continue
    name = d.rsplit(".", 1)[0]
    acc = res.setdefault(name, {})
    tmp = {}
    for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "solve_kernel" not in row["Kernel_Name"]:
                continue
            a = tmp.setdefault(row["Counter_Name"], [0.0, set()]); a[0] += float(row["Counter_Value"]); a[1].add(row.get("Dispatch_Id"))
    for c, v in tmp.items():
        acc[c] = v[0] / max(len(v[1]), 1)
full = res.get("full", {})
print("per launch of the solve kernel (B = 1024, 8 fixed iterations); WRITE_SIZE / FETCH_SIZE raw KiB counters x 1024 -> MB; instruction counters in millions")
print("%-12s" % "without" + "".join("%18s" % c.replace("SQ_INSTS_", "") for c in CNT))
def fmt(c, v): return v * 1024 / 1e6 if c.endswith("_SIZE") else v / 1e6
for d, r in res.items():
    print("%-12s" % d + "".join("%18.1f" % fmt(c, r.get(c, float("nan"))) for c in CNT))
print("\nphase = full - without:")
for d, r in res.items():
    if d != "full":
        print("%-12s" % d + "".join("%18.1f" % fmt(c, full.get(c, float("nan")) - r.get(c, float("nan"))) for c in CNT))
PY
cat $R/gpurun_out/abl_wr_table.txt
rm -rf $O/*/
