#!/bin/bash
# Same-device A/B of (objects per lane, stage capacity) of the band kernel:  tools/try_caps.sh <label> [bench args...]
L=$1; shift
for rc in "2 192" "2 288" "4 288"; do
  set -- "$@"
  r=${rc% *}; c=${rc#* }
  python bench.py "$@" --cpu-seconds 0 --set tile_r=$r --set band_cap=$c 2>/dev/null | python tools/bench_line.py "$L R=$r cap=$c"
done
