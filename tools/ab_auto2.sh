#!/bin/bash
# same-box A/B of tools/probe_auto_prof.py (DD, DR, RR of config #4 on the AUTO path) over library variants and context options:
#   tools/ab_auto2.sh <name> "<variant>|<YAW_SET string>" ...    ("-" or empty variant = the product library)
NAME=$1; shift
OUT=gpurun_out/abauto2_$NAME.txt; mkdir -p gpurun_out; : > $OUT
for V in "$@"; do
  TAG="${V%%|*}"; SET="${V#*|}"; [ "$SET" == "$V" ] && SET=""
  LIB=""; [ -n "$TAG" ] && [ "$TAG" != "-" ] && LIB="$PWD/yet_another_wizz_amd/build/variants/libyawhip_$TAG.so"
  echo "== [$TAG|$SET]" >> $OUT
  YAW_SET="$SET" YAW_AMD_LIB=$LIB timeout -k 10 300 python tools/probe_auto_prof.py ${AUTO_ARGS:-1e7 1e8 w} 2>&1 | grep -E "^(DD|DR|RR):|rror" | cut -c1-200 >> $OUT
done
cat $OUT
