#!/bin/bash
# same-box A/B of tools/probe_auto.py over library variants: tools/ab_auto.sh <name> <variant> ...
NAME=$1; shift
OUT=gpurun_out/abauto_$NAME.txt; : > $OUT
for TAG in "$@"; do
  LIB=""; [ "$TAG" != "-" ] && LIB="$PWD/yet_another_wizz_amd/build/variants/libyawhip_$TAG.so"
  echo "== $TAG" >> $OUT
  YAW_AMD_LIB=$LIB timeout -k 10 300 python tools/probe_auto.py 1e7 1e8 w 2>&1 | grep -E "auto->|band->" >> $OUT
done
cat $OUT
