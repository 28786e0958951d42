"""Quick throughput probe of the raw C-ABI path (not the bench): N x N uniform box, P patches, B bins."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from yet_another_wizz_amd import _lib

def make(rng, n, P, nb, side):
    ra = np.deg2rad(rng.uniform(0, side, n)); dec = np.arcsin(rng.uniform(0, np.sin(np.deg2rad(side)), n))
    x = np.cos(ra)*np.cos(dec); y = np.sin(ra)*np.cos(dec); z = np.sin(dec)
    g = int(np.sqrt(P)); pi = np.minimum((ra/np.deg2rad(side)*g).astype(int), g-1)*g + np.minimum((np.sin(dec)/np.sin(np.deg2rad(side))*g).astype(int), g-1)
    k = rng.integers(0, nb, n)
    key = pi*nb + k; o = np.argsort(key, kind="stable")
    off = np.zeros(P*nb+1, dtype=np.int64); np.cumsum(np.bincount(key, minlength=P*nb), out=off[1:])
    return x[o], y[o], z[o], off

def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
    P, B = 16, 30
    rng = np.random.default_rng(1)
    ctx = _lib.Context(0)
    x, y, z, off = make(rng, n, P, B, 30.0); c1 = _lib.DeviceCatalog(ctx, x, y, z, None, P, B, off)
    x, y, z, off = make(rng, n, P, 1, 30.0); c2 = _lib.DeviceCatalog(ctx, x, y, z, None, P, 1, off)
    jobs = np.array([(p, q) for p in range(P) for q in range(P)], dtype=np.int32)[: int(sys.argv[2]) if len(sys.argv) > 2 else 64]
    am = np.pi/10800
    t = np.tile(np.array([(2*np.sin(am/2))**2, (2*np.sin(10*am/2))**2]), (B, 1))
    import os
    if os.environ.get("NO_HITS"): ctx.set_option("debug_no_hits", 1)
    for kern in sys.argv[3:] or ["exact"]:
        for r in (1, 2, 4):
            ctx.set_option("tile_r", r)
            for rep in range(2):
                counts, _, st = _lib.count_pairs(ctx, c1, c2, jobs, t, kernel=kern)
            print(f"{kern} R={r} cand={st.candidate_pairs:.3e} kernel_ms={st.kernel_ms:.2f} total_ms={st.total_ms:.2f} "
                  f"rate={st.candidate_pairs/st.kernel_ms/1e6:.1f} Gpairs/s eval={st.evaluated_pairs:.3e} evalrate={st.evaluated_pairs/st.kernel_ms/1e6:.1f} G/s found={counts.sum()} wgs={st.n_workgroups}", flush=True)

main()
