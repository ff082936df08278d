#!/bin/bash
# GPU timeline of the 128-stream replay (4 host threads): how busy the device is, what overlaps, how long a launch waits
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05z45; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace -d $O/tr -o tl --output-format csv -- python3 $R/bench.py --mode replay --streams 128 --steps 30 --warmup 8 --no-cpu-baseline > $O/log.txt 2>&1
tail -c 400 $O/log.txt | head -c 300; echo
python3 - <<PY
import csv, glob, collections
O="$O"
K=[]
for f in glob.glob(O+"/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        K.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:34], r.get("Queue_Id","")))
Cp=[]
for f in glob.glob(O+"/tr/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        Cp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction","")))
K.sort()
sol=[k for k in K if "solve_kernel" in k[2]]
t0, t1 = sol[len(sol)//3][0], sol[-1][1]      # the steady part
W=[k for k in K if k[0]>=t0 and k[1]<=t1]
# union of busy intervals
ev=sorted([(a,1) for a,b,_,_ in W]+[(b,-1) for a,b,_,_ in W])
busy=0; depth=0; last=None; depth_time=collections.Counter()
for t,d in ev:
    if last is not None: depth_time[depth]+=t-last
    depth+=d; last=t
tot=t1-t0
print("window %.1f ms: some kernel running %.1f %% of the time; time by number of kernels in flight:" % (tot/1e6, 100*(tot-depth_time[0])/tot), {k: "%.1f %%" % (100*v/tot) for k,v in sorted(depth_time.items())})
by=collections.defaultdict(list)
for a,b,n,q in W: by[n].append((b-a)/1e3)
for n,v in sorted(by.items(), key=lambda kv:-sum(kv[1])):
    print("  %-36s %5d launches, mean %8.1f us, sum %7.1f ms (%.1f %% of the window)" % (n, len(v), sum(v)/len(v), sum(v)/1e3, 100*sum(v)*1e3/tot))
cps=[c for c in Cp if c[0]>=t0 and c[1]<=t1]
byc=collections.defaultdict(list)
for a,b,d in cps: byc[d].append((b-a)/1e3)
for d,v in byc.items(): print("  copies %-28s %5d, mean %7.1f us, sum %6.1f ms" % (d, len(v), sum(v)/len(v), sum(v)/1e3))
# per solve: how long since the last H2D copy that ended before it on any queue (queueing delay proxy): gap between the big upload's end and the solve's start
big=[c for c in cps if "HOST_TO_DEVICE" in c[2] and (c[1]-c[0])>30e3]
import bisect
ends=sorted(c[1] for c in big)
gaps=[]
for s in sol:
    if s[0]<t0: continue
    i=bisect.bisect_right(ends, s[0])-1
    if i>=0: gaps.append((s[0]-ends[i])/1e3)
gaps.sort()
print("solve start minus the end of the latest large upload before it: median %.0f us, 90 %% %.0f us, max %.0f us" % (gaps[len(gaps)//2], gaps[int(0.9*len(gaps))], gaps[-1]))
PY
