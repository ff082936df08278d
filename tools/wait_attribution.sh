#!/bin/bash
# Round 6, review item 1(a): WHERE the solve kernel's wavefronts wait.  Separate --pmc passes (kernel-trace only, as gpurun requires) of
# the benchmark command; every pass is allowed to fail (an unknown counter name ends that pass only), the table is made from what came back.
#   bash tools/wait_attribution.sh [tag] [lib]      ->  gpurun_out/wait_<tag>/{table.txt,counters_avail.txt,*.csv}
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/wait_$TAG
mkdir -p $O
[ -n "$2" ] && export TCV_LIB=$R/$2
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_avail_full.txt 2>&1
grep -o -E "\b(SQ|SQC|TCP|TCC|TA|TD|GRBM|SPI)_[A-Z0-9_a-z]+" $O/counters_avail_full.txt | sort -u > $O/counters_avail.txt
CMD="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
pass() {      # name, counters...
  N=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $O/$N -o p --output-format csv -- $CMD > $O/$N.log 2>&1
  echo "$N rc $?" >> $O/passes.txt
}
pass wait   SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_IFETCH
pass level  SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_IFETCH_LEVEL SQ_WAVE_CYCLES
pass active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_INSTS_SALU
pass icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
pass dcache SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE
pass vmem   SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_VALU SQ_WAVE_CYCLES
pass vmem2  SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT SQ_INSTS_VALU SQ_WAVE_CYCLES
pass tcp    TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum
pass tcp2   TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass tcc    TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass lds    SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES
pass mfma   SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_INSTS_VALU
python3 - <<PY > $O/table.txt 2>&1
import csv, glob, collections
O = "$O"
acc = collections.defaultdict(lambda: collections.defaultdict(dict))
for d in sorted(glob.glob(O + "/*/")):
    name = d.rstrip("/").split("/")[-1]
    tmp = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "tcv::" in r["Kernel_Name"]:
                tmp[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in tmp.items():
        for c, v in cs.items():
            acc[k][name][c] = sum(v) / len(v)
for k in sorted(acc):
    if "solve_kernel" not in k and "marg_kernel" not in k:
        continue
    print("==", k)
    for name in acc[k]:
        print("  [%s]" % name)
        for c, v in sorted(acc[k][name].items()):
            print("    %-34s %.6g" % (c, v))
PY
cat $O/passes.txt
rm -rf $O/*/
head -150 $O/table.txt
