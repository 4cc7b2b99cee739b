# PMC passes over the split Delta kernels (run on the GPU box): bash tools/experiments/pmc_c1.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-x}
i=0
for ctrs in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAIT_INST_LDS" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs -d $R/gpurun_out/pmc_${T}_$i -o x -- python $R/tools/experiments/c1_variants.py c1=0 > $R/gpurun_out/pmc_${T}_$i.log 2>&1
done
