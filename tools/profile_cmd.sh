#!/bin/bash
# rocprofv3 evidence for any python probe (not only bench.py) on the GPU box:
#   tools/profile_cmd.sh <tag> <script.py> [args...]
# -> gpurun_out/prof_<tag>/{stats,pmc_fetch,pmc_write,sq1,sq2,sq3}; counters in their own runs with --kernel-trace only.
set -eo pipefail
TAG=$1; shift
SCRIPT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$REPO/$SCRIPT" "$@" > "$OUT/stats.log" 2> "$OUT/stats.err"
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o run -- python3 "$REPO/$SCRIPT" "$@" > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o run -- python3 "$REPO/$SCRIPT" "$@" > /dev/null 2> "$OUT/write.err"
echo "pmc passes done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/sq1" -o run -- python3 "$REPO/$SCRIPT" "$@" > /dev/null 2> "$OUT/sq1.err"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM --kernel-trace --output-format csv -d "$OUT/sq2" -o run -- python3 "$REPO/$SCRIPT" "$@" > /dev/null 2> "$OUT/sq2.err"
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_CYCLES --kernel-trace --output-format csv -d "$OUT/sq3" -o run -- python3 "$REPO/$SCRIPT" "$@" > /dev/null 2> "$OUT/sq3.err" || echo "sq3 pass failed (counter names)"
echo "sq passes done"
find "$OUT" -name "*.csv" | wc -l
