# A/B of compile-time variants (and run-time options) on ONE device (timings from different boxes differ by ~10 %)
#   tools/try_variants.sh "-DYAW_MSTAGE=64|--strip-micro 5000 --tile-r 1" "-DYAW_MSTAGE=128|..." ...   (YAW_MWG is fixed at 64)
set -e
cd ${GRAFT_REPO_ROOT:-.}
# whatever happens, leave the default build behind (build.py also refuses to treat a library built with other flags as fresh)
trap 'python -c "from yet_another_wizz_amd import build; build.build_library(force=True)"' EXIT
for round in 1 2; do
for V in "$@"; do
  FL="${V%%|*}"; OPT=""
  if [[ "$V" == *"|"* ]]; then OPT="${V#*|}"; fi
  python -c "
from yet_another_wizz_amd import build
build.build_library(force=True, extra_flags='$FL'.split())"
  python bench.py --steps 4 --warmup 1 --cpu-seconds 0 $OPT 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V', 'round $round', round(d['ms_per_step'],2), round(d['kernel_ms_per_step'],2), '%.3e' % d['evaluated_pairs_per_step'])"
done
done
python -c "
from yet_another_wizz_amd import build
build.build_library(force=True)"
