#!/bin/bash
# LDS-focused PMC pass over the Delta-kernel ablation binary (run on the GPU box): tools/pmc_ablate.sh <tag>
set -u
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_abl_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL -d "$OUT/p1" -o b -- $ROOT/tools/bin/delta_j2_ablate > "$OUT/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_CMD_FIFO_FULL -d "$OUT/p2" -o b -- $ROOT/tools/bin/delta_j2_ablate > "$OUT/p2.log" 2>&1
tail -3 "$OUT"/p*.log
