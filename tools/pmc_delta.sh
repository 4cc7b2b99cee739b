#!/bin/bash
# PMC passes focused on the Delta kernel (run on the GPU box).  tools/pmc_delta.sh <tag> [bench args]
set -u
TAG=${1:-x}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --accuracy-pairs 0 $*"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES -d "$OUT/p1" -o b -- $BENCH > "$OUT/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD -d "$OUT/p2" -o b -- $BENCH > "$OUT/p2.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_IFETCH SQ_WAVES GRBM_GUI_ACTIVE -d "$OUT/p3" -o b -- $BENCH > "$OUT/p3.log" 2>&1
tail -2 "$OUT"/p*.log
