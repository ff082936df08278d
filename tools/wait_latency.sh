#!/bin/bash
# Round 6, review item 1(a), second half: mean latencies (rocprofv3's accumulate() derived counters = Little's law on the SQ level counters)
# and L2 hit rates of the solve kernel -- for the production launch shape and for the four configurations of the occupancy-3
# experiment (tools/dev_occupancy3.py; profiles/r04_occupancy3.txt: why does a third workgroup slow its neighbours 1.72 x?).
#   bash tools/wait_latency.sh [tag]   ->  gpurun_out/lat_<tag>/table.txt
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lat_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PASSES=("vmemlat:VmemLatency" "ldslat:LdsLatency" "smemlat:SmemLatency" "iflat:InstrFetchLatency" "occ:MeanOccupancyPerActiveCU"
        "wait:SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM"
        "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "tcprd:TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "tcpwr:TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"
        "tcplat:TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" "fetch:FETCH_SIZE" "write:WRITE_SIZE" "vm:SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE")
run_cfg() {      # name, command...
  CFG=$1; shift
  for P in "${PASSES[@]}"; do
    N=${P%%:*}; CTR=${P#*:}
    timeout 600 rocprofv3 --kernel-trace --pmc $CTR -d $O/$CFG.$N -o p --output-format csv -- "$@" > $O/$CFG.$N.log 2>&1
    echo "$CFG $N rc $?" >> $O/passes.txt
  done
}
unset TCV_LIB TCV_GRID TCV_CHAIN_LDS_DOUBLES
run_cfg bench python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras
export TCV_CHAIN_LDS_DOUBLES=6784
export TCV_LIB=$R/tc-viml_amd/libtcv_hip.so TCV_GRID=512
run_cfg occA python3 $R/tools/dev_occupancy3.py --worker 6 1536
export TCV_LIB=$R/tc-viml_amd/libtcv_hip_occ3.so TCV_GRID=512
run_cfg occB python3 $R/tools/dev_occupancy3.py --worker 6 1536
export TCV_LIB=$R/tc-viml_amd/libtcv_hip_occ3.so TCV_GRID=768
run_cfg occC python3 $R/tools/dev_occupancy3.py --worker 6 1536
unset TCV_CHAIN_LDS_DOUBLES
export TCV_LIB=$R/tc-viml_amd/libtcv_hip.so TCV_GRID=256
run_cfg g256 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras
export TCV_LIB=$R/tc-viml_amd/libtcv_hip_occ1.so TCV_GRID=256
run_cfg occ1 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras
unset TCV_LIB TCV_GRID TCV_CHAIN_LDS_DOUBLES
python3 - <<PY > $O/table.txt 2>&1
import csv, glob, collections
O = "$O"
acc = collections.defaultdict(lambda: collections.defaultdict(dict))
dur = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(O + "/*/")):
    cfg, name = d.rstrip("/").split("/")[-1].split(".")
    tmp = collections.defaultdict(list)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "solve_kernel" in r["Kernel_Name"]:
                tmp[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in tmp.items():
        acc[cfg][name][c] = (sum(v) / len(v), len(v))
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "solve_kernel" in r["Kernel_Name"]:
                dur[cfg][name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for cfg in acc:
    print("==", cfg)
    for name in acc[cfg]:
        ds = dur[cfg].get(name, [])
        print("  [%s]  kernel ms (under the counters): %s" % (name, ", ".join("%.3f" % x for x in ds[-4:])))
        for c, (v, n) in sorted(acc[cfg][name].items()):
            print("    %-34s %.6g   (%d launches)" % (c, v, n))
PY
cat $O/passes.txt | grep -v "rc 0"
rm -rf $O/*/
cat $O/table.txt
