#!/bin/bash
# Round 3, review item 4 (i): what do the 156 spilled VGPRs of the production chain kernel cost?  A/B at EQUAL occupancy (one workgroup per
# CU: TCV_GRID=256) between the production build (256-register budget, 156 spills, 1072 B of scratch per lane) and the same source compiled
# for one wavefront per SIMD (libtcv_hip_occ1.so: 370 registers, 8 spills, 684 B): kernel time and HBM write traffic (WRITE_SIZE).
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/spill; mkdir -p $O; cd $R
for lib in libtcv_hip.so libtcv_hip_occ1.so; do
  TCV_LIB=$R/tc-viml_amd/$lib TCV_GRID=256 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_$lib.json 2>/dev/null
  python3 -c "import json; d=json.load(open('$O/bench_$lib.json')); print('$lib', 'TCV_GRID=256 solve kernel ms', round(d['kernel_ms']['solve'],3), 'marg', round(d['kernel_ms']['marginalize'],3))"
done
cd /tmp && export TMPDIR=/tmp
for lib in libtcv_hip.so libtcv_hip_occ1.so; do
  TCV_LIB=$R/tc-viml_amd/$lib TCV_GRID=256 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w_$lib -o w --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  TCV_LIB=$R/tc-viml_amd/$lib TCV_GRID=256 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f_$lib -o f --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
done
cd $R
python3 - <<PY
import csv, glob, collections
for lib in ("libtcv_hip.so", "libtcv_hip_occ1.so"):
    for tag, ctr in (("w", "WRITE_SIZE"), ("f", "FETCH_SIZE")):
        acc = collections.defaultdict(list)
        for f in glob.glob("$O/%s_%s/**/*counter_collection.csv" % (tag, lib), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr and "solve_kernel" in r["Kernel_Name"]:
                    acc[r["Kernel_Name"][:50]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(lib, ctr, k, "launches", len(v), "mean per launch: %.3f GB (KiB counter x 1024)" % (sum(v) / len(v) * 1024 / 1e9))
PY
