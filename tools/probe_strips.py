"""Experiment: strips as virtual patches, entirely on the host. Each patch is cut along a second axis into
strips of a global grid; (patch, strip) becomes a virtual patch and a job (p, q) becomes the sub-jobs between
strips at most `reach` apart. The library is unchanged. Prints kernel time and evaluated pairs vs the plain layout."""
import sys, time, types
import numpy as np
sys.path.insert(0, ".")
import bench
import yet_another_wizz_amd as yaw
from yet_another_wizz_amd import engine, _lib
from yet_another_wizz_amd.measurements import angular_plans, threshold_table

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
delta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
args = types.SimpleNamespace(n_ref=n, n_unk=n, patches=64, zbins=30)
config, ref, unk = bench.make_catalogs(args)
lref = ref.build_trees(config.binning.edges, closed=config.binning.closed)
lunk = unk.build_trees(None)
links = yaw.PatchLinkage.from_catalogs(config, ref, unk)
jobs = links.get_patch_pairs(ref, unk)
t = threshold_table(angular_plans(config))
ctx = engine.get_context()
fine0, st0 = engine.count_fine(lref, lunk, jobs, t)
fine0, st0 = engine.count_fine(lref, lunk, jobs, t)
print(f"plain: kernel {st0.kernel_ms:.2f} ms total {st0.total_ms:.2f} eval {st0.evaluated_pairs:.3e} wgs {st0.n_workgroups}")

def virtual(layout, nb, axis_col, delta):
    """re-segment a PatchLayout into (patch, strip) virtual patches on the global strip grid"""
    P = layout.num_patches
    ns = int(np.ceil(2.0 / delta))
    off = layout.offsets
    seg_of = np.repeat(np.arange(P * nb), np.diff(off))
    patch_of, bin_of = seg_of // nb, seg_of % nb
    v = getattr(layout, axis_col)
    gs = np.clip(np.floor((v + 1.0) / delta).astype(np.int64), 0, ns - 1)
    key = (patch_of * ns + gs) * nb + bin_of
    order = np.argsort(key, kind="stable")
    noff = np.zeros(P * ns * nb + 1, dtype=np.int64)
    np.cumsum(np.bincount(key, minlength=P * ns * nb), out=noff[1:])
    dev = _lib.DeviceCatalog(ctx, layout.x[order], layout.y[order], layout.z[order], None, P * ns, nb, noff)
    occupied = np.unique(patch_of * ns + gs)
    return dev, ns, occupied

for delta in (delta, delta / 2):
    t0 = time.perf_counter()
    d1, ns, occ1 = virtual(lref, 30, "y", delta)
    d2, _, occ2 = virtual(lunk, 1, "y", delta)
    reach = int(np.ceil(np.sqrt(t.max()) / delta))
    occ1s, occ2s = set(occ1.tolist()), set(occ2.tolist())
    sub, parent = [], []
    strips2 = {}
    for vp in occ2: strips2.setdefault(vp // ns, []).append(vp % ns)
    for j, (p, q) in enumerate(jobs):
        for s2 in strips2.get(q, []):
            for ds in range(-reach, reach + 1):
                if p * ns + s2 + ds in occ1s:
                    sub.append((p * ns + s2 + ds, q * ns + s2)); parent.append(j)
    sub = np.array(sub, dtype=np.int32); parent = np.array(parent)
    prep = time.perf_counter() - t0
    for rep in range(2):
        counts, _, st = _lib.count_pairs(ctx, d1, d2, sub, t, kernel="sweep")
    tot = np.zeros_like(fine0)
    np.add.at(tot, parent, counts.astype(np.float64))
    print(f"strips {delta}: sub-jobs {len(sub)} reach {reach} kernel {st.kernel_ms:.2f} ms total {st.total_ms:.2f} ms eval {st.evaluated_pairs:.3e} "
          f"wgs {st.n_workgroups} equal={np.array_equal(tot, fine0)} prep {prep:.1f}s", flush=True)
    d1.free(); d2.free()
