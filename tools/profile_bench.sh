#!/bin/bash
# rocprofv3 passes over the default bench command (run on the GPU box through gpurun).
#   tools/profile_bench.sh <tag>        -> gpurun_out/prof_<tag>/{stats,pmc_fetch,pmc_write,pmc_mfma,pmc_clk}/...
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-walk-stats --traffic none ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench -- $BENCH > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o bench -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o bench -- $BENCH > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA -d "$OUT/pmc_mfma" -o bench -- $BENCH > "$OUT/pmc_mfma.log" 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d "$OUT/pmc_clk" -o bench -- $BENCH > "$OUT/pmc_clk.log" 2>&1   # engine cycles (sum over the 8 XCDs) -> clock under load
find "$OUT" -name "*.csv" | head -40
for f in $(find "$OUT/stats" -name "*kernel_stats.csv"); do echo "== $f"; head -12 "$f"; done
