# A/B of compile-time variants on ONE device (timings from different boxes differ by up to ~10 %)
#   tools/try_variants.sh "-DYAW_CERTAIN=0" "-DYAW_CERTAIN=1" ...
set -e
cd ${GRAFT_REPO_ROOT:-.}
for round in 1 2; do
for FL in "$@"; do
  python -c "
from yet_another_wizz_amd import build
build.build_library(force=True, extra_flags='$FL'.split())"
  python bench.py --steps 4 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$FL', 'round $round', round(d['ms_per_step'],2), round(d['kernel_ms_per_step'],2))"
done
done
python -c "
from yet_another_wizz_amd import build
build.build_library(force=True)"
