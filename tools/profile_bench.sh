#!/bin/bash
# Collect the rocprofv3 evidence for one bench configuration on the GPU box.
#   tools/profile_bench.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{stats,pmc_fetch,pmc_write,sq1,sq2}/... ; tools/summarize_profile.py condenses them into profiles/.
# (counters in their own runs with --kernel-trace only: the pool refuses --pmc together with the API trace domains)
set -eo pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# 1) kernel trace + stats (timing)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$REPO/bench.py" --cpu-seconds 0 "$@" > "$OUT/bench_stats.json" 2> "$OUT/bench_stats.err"
# 2) PMC passes, one counter group per run (FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o run -- python3 "$REPO/bench.py" --cpu-seconds 0 "$@" > "$OUT/bench_fetch.json" 2> "$OUT/bench_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o run -- python3 "$REPO/bench.py" --cpu-seconds 0 "$@" > "$OUT/bench_write.json" 2> "$OUT/bench_write.err"
# 3) SQ counters of the count kernel, two passes of eight
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/sq1" -o run -- python3 "$REPO/bench.py" --cpu-seconds 0 "$@" > /dev/null 2> "$OUT/sq1.err"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM --kernel-trace --output-format csv -d "$OUT/sq2" -o run -- python3 "$REPO/bench.py" --cpu-seconds 0 "$@" > /dev/null 2> "$OUT/sq2.err"
find "$OUT" -name "*.csv" | wc -l
