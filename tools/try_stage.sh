set -e
cd $GRAFT_REPO_ROOT
for ST in 256 128 64; do
  python -c "
from yet_another_wizz_amd import build
build.build_library(force=True, extra_flags=['-DYAW_MSTAGE=$ST'])"
  python bench.py --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('MSTAGE=$ST', d['ms_per_step'], d['kernel_ms_per_step'])"
done
