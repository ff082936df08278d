"""Table of tools/dev_ablate_pmc.sh: per-launch counters of the solve kernel per removed phase and the difference to the full run."""
import csv, glob, os, sys
root = sys.argv[1]
CNT = ["SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_SALU", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY"]
res = {}
for d in sorted(os.listdir(root)):
    if not os.path.isdir(os.path.join(root, d)):
        continue
    acc = {c: [0.0, set()] for c in CNT}
    for f in glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "solve_kernel" not in row["Kernel_Name"] or row["Counter_Name"] not in acc:
                continue
            a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1].add(row.get("Dispatch_Id"))
    res[d] = {c: (v[0] / max(len(v[1]), 1)) for c, v in acc.items()}
full = res.get("full")
print("counters per launch of the solve kernel (B = 1024, every step accepted), in millions; 'phase' = full - without")
print("%-12s" % "without" + "".join("%22s" % c.replace("SQ_", "") for c in CNT) + "   conflict/active")
for d, r in res.items():
    print("%-12s" % d + "".join("%22.2f" % (r[c] / 1e6) for c in CNT) + "   %.3f" % (r["SQ_LDS_BANK_CONFLICT"] / max(r["SQ_ACTIVE_INST_LDS"], 1)))
if full:
    print("\nphase = full - without:")
    for d, r in res.items():
        if d == "full":
            continue
        dl = {c: full[c] - r[c] for c in CNT}
        print("%-12s" % d + "".join("%22.2f" % (dl[c] / 1e6) for c in CNT) + "   %.3f" % (dl["SQ_LDS_BANK_CONFLICT"] / dl["SQ_ACTIVE_INST_LDS"] if abs(dl["SQ_ACTIVE_INST_LDS"]) > 1 else 0.0))
