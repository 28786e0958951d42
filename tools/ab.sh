#!/bin/bash
# Same-box A/B of bench.py runs (boxes differ by ~10 %):
#   tools/ab.sh <name> "<variant>|<bench args>" ...      -> gpurun_out/ab_<name>.txt
# <variant> = tag of a library built by tools/build_variant.py (empty = the product library).
NAME=$1; shift
OUT=gpurun_out/ab_$NAME.txt
mkdir -p gpurun_out; : > $OUT
for V in "$@"; do
  TAG="${V%%|*}"; ARGS="${V#*|}"
  LIB=""; [ -n "$TAG" ] && LIB="$PWD/yet_another_wizz_amd/build/variants/libyawhip_$TAG.so"
  YAW_AMD_LIB=$LIB timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 $ARGS 2>/dev/null | python tools/bench_line.py "[$TAG|$ARGS]" >> $OUT
done
cat $OUT
