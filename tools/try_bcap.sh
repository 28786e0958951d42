#!/bin/bash
# Same-device A/B of the small stage capacity of the band kernel (compile-time YAW_BCAP):  tools/try_bcap.sh 160 192 224
cd ${GRAFT_REPO_ROOT:-.}
trap 'python -c "from yet_another_wizz_amd import build; build.build_library(force=True)" > /dev/null' EXIT
for C in "$@"; do
  python -c "
from yet_another_wizz_amd import build
build.build_library(force=True, extra_flags=['-DYAW_BCAP=$C'])" > /dev/null 2>&1
  python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --set band_cap=$C 2>/dev/null | python tools/bench_line.py "bcap=$C 10M"
  python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --weights --set band_cap=$C 2>/dev/null | python tools/bench_line.py "bcap=$C 10Mw"
  python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --n-ref 2e7 --n-unk 2e7 --set band_cap=$C 2>/dev/null | python tools/bench_line.py "bcap=$C 20M"
done
