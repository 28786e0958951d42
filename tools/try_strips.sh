# A/B of run-time options on ONE device: strip grid spacing x tile size
set -e
cd ${GRAFT_REPO_ROOT:-.}
for round in 1 2; do
for OPT in "$@"; do
  python bench.py --steps 4 --warmup 1 --cpu-seconds 0 $OPT 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$OPT', 'round $round', round(d['ms_per_step'],2), round(d['kernel_ms_per_step'],2), '%.3e' % d['evaluated_pairs_per_step'])"
done
done
