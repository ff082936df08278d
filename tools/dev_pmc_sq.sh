cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for V in 0 1; do
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $R/gpurun_out/pmc_sq_v$V -o sq --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --variant $V > $R/gpurun_out/pmc_sq_v$V.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA -d $R/gpurun_out/pmc_sq2_v$V -o sq --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --variant $V > $R/gpurun_out/pmc_sq2_v$V.log 2>&1
done
ls -R $R/gpurun_out/pmc_sq_v0 | head
