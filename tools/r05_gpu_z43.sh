#!/bin/bash
# GPU timeline of a lock-step frame: kernels and copies with their time stamps (one host thread, 8 streams = 8 windows per call)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05z43; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace -d $O/tr -o tl --output-format csv -- python3 $R/bench.py --mode replay --streams 8 --steps 40 --warmup 10 --host-threads 1 --no-cpu-baseline > $O/log.txt 2>&1
ls $O/tr/* | head
python3 - <<PY
import csv, glob
O="$O"
ev=[]
for f in glob.glob(O+"/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K:"+r["Kernel_Name"][:40]))
for f in glob.glob(O+"/tr/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C:"+r.get("Direction", r.get("Kind",""))+":"+r.get("Size","")))
ev.sort()
# find the last 6 solve kernels and print the events around each (from the previous solve's end to this solve's end + 0.4 ms)
sol=[i for i,e in enumerate(ev) if "solve_kernel" in e[2]]
print("events", len(ev), "solve launches", len(sol))
for si in sol[-4:-1]:
    t0=ev[si][0]
    print("---- frame: times in us relative to the solve kernel's start")
    for e in ev:
        if t0-700e3 < e[0] < ev[si][1]+900e3:
            print("  %9.1f .. %9.1f  (%7.1f us)  %s" % ((e[0]-t0)/1e3, (e[1]-t0)/1e3, (e[1]-e[0])/1e3, e[2]))
PY
