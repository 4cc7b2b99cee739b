#!/bin/bash
# L2 / L1 cache counters for the head kernels.  tools/pmc_cache.sh <tag> [bench args]
set -u
TAG=${1:-x}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --accuracy-pairs 0 $*"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d "$OUT/c1" -o b -- $BENCH > "$OUT/c1.log" 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum -d "$OUT/c2" -o b -- $BENCH > "$OUT/c2.log" 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -d "$OUT/c3" -o b -- $BENCH > "$OUT/c3.log" 2>&1
grep -ciE "error|invalid" "$OUT"/c*.log
