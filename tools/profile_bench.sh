#!/bin/bash
# Collect the rocprofv3 evidence for one bench configuration on the GPU box.
#   tools/profile_bench.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{stats,pmc_fetch,pmc_write}/... (copy the summaries into profiles/).
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
find "$OUT" -name "*.csv" | head -30
