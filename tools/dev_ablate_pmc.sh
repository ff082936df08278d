# hardware counters of the solve kernel with one phase removed at a time (ablation build): differences to the full run attribute
# LDS instructions / bank-conflict cycles / SALU and VALU instructions to the phases.  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export TCV_LIB=$R/tc-viml_amd/libtcv_hip_abl.so
O=$R/gpurun_out/abl_pmc
mkdir -p $O
# name:mask  (bits as in tools/dev_ablate.py)
for NM in full:0 vis_eval:0x1 vis_gather:0x2 schur:0x8 prior_A:0x10 imu_raw:0x20 imu_gather:0x80 fin_scale:0x200 chain_fwd:0x800 chol:0x1000 back:0x2000 ch_T:0x100000 ch_owners:0x200000 ch_mfma:0x400000 everything:0x7ffff; do
  N=${NM%%:*}; M=${NM##*:}
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $O/$N -o p --output-format csv -- python3 $R/tools/dev_ablate_one.py $M 1024 3 > $O/$N.log 2>&1
  tail -1 $O/$N.log
done
python3 $R/tools/dev_ablate_pmc_table.py $O > $R/gpurun_out/abl_pmc_table.txt 2>&1
cat $R/gpurun_out/abl_pmc_table.txt
rm -rf $O/*/  # the raw csv trees are large
