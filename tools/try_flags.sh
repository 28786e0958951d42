#!/bin/bash
# Same-device A/B of compile-time flags over a few workloads:  tools/try_flags.sh "" "-DYAW_BAND_TAILS=0" ...
cd ${GRAFT_REPO_ROOT:-.}
trap 'python -c "from yet_another_wizz_amd import build; build.build_library(force=True)" > /dev/null' EXIT
for F in "$@"; do
  python -c "
from yet_another_wizz_amd import build
build.build_library(force=True, extra_flags='$F'.split())" > /dev/null 2>&1
  for a in "--steps 20 --warmup 5" "--weights --steps 20 --warmup 5" "--n-ref 1e6 --n-unk 1e6 --steps 50 --warmup 10" "--n-ref 2.5e7 --n-unk 2.5e7 --steps 5 --warmup 2" "--n-ref 5e7 --n-unk 5e7 --patches 128 --scales 3 --steps 3 --warmup 1"; do
    python bench.py $a --cpu-seconds 0 2>/dev/null | python tools/bench_line.py "[$F] $a"
  done
  if [ -n "$PROBE_AUTO" ]; then python tools/probe_auto.py 1e7 1e8 w 2>&1 | grep -E "RR auto:|kernels:|end to end"; fi
done
