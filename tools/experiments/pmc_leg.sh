# PMC passes over the batched leg (run on the GPU box): bash tools/experiments/pmc_leg.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-leg}
i=0
for ctrs in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAIT_INST_LDS" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs -d $R/gpurun_out/pmc_${T}_$i -o x -- python $R/tools/experiments/leg_time.py > $R/gpurun_out/pmc_${T}_$i.log 2>&1
done
