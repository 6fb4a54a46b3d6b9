export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
rm -f $O/pmc_final_two_role.txt
for pat in interact_bwd_members node_interact_fwd_grouped node_interact_weight; do
  echo "== $pat" >> $O/pmc_final_two_role.txt
  KBENCH_OPS=layer bash tools/pmc_kernel.sh $pat \
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
    "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
    "TCC_HIT_sum TCC_MISS_sum SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH" >> $O/pmc_final_two_role.txt 2>&1
done
cat $O/pmc_final_two_role.txt
