import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from oracle import oracle
from yet_another_wizz_amd import _lib
from test_gpu_kernel_parity import _random_catalog, _upload

ctx = _lib.Context(0)
rng = np.random.default_rng(99 + 1)
P, B = 6, 5
c1 = _random_catalog(rng, 7000, P, B, False)
c2 = _random_catalog(rng, 9000, P, 1, False)
jobs = np.array([(p, q) for p in range(P) for q in range(P) if (p + q) % 3 != 1], dtype=np.int32)
for ne in (2, 4):
    if ne == 4:
        lim = oracle.parse_ang_limits(np.array([0.5, 2.0]) * np.pi / 10800, np.array([3.0, 8.0]) * np.pi / 10800)
    else:
        lim = oracle.parse_ang_limits(np.array([0.5]) * np.pi / 10800, np.array([8.0]) * np.pi / 10800)
    for uni in (True, False):
        t = np.stack([oracle.thresholds_for(oracle.ang_bins_for(lim * (1.0 + (0.0 if uni else 0.1) * k), None, None)) for k in range(B)])
        d1, d2 = _upload(ctx, c1), _upload(ctx, c2)
        exp_c, _ = oracle.count_jobs(c1, c2, jobs, t)
        for tile_r in (2, 1, 4):
            ctx.set_option("tile_r", tile_r)
            counts, _, st = _lib.count_pairs(ctx, d1, d2, jobs, t, kernel="band", want_counts=True, want_sums=False)
            bad = np.argwhere(counts != exp_c)
            print(f"ne={ne} uni={uni} R={tile_r} kernel={st.kernel_used} mode={st.layout_mode} eval={st.evaluated_pairs} "
                  f"sum got={counts.sum()} exp={exp_c.sum()} mismatched cells={len(bad)}", flush=True)
            if len(bad):
                for b in bad[:6]:
                    print("   job", jobs[b[0]], "bin", b[1], "fine", b[2], "got", counts[tuple(b)], "exp", exp_c[tuple(b)])
