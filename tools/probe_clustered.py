"""Timing of the clustered survey (tests/clustered_sky.py) on the GPU: cross count (weighted) and autocorrelation count."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import clustered_sky as cs
import yet_another_wizz_amd as yaw
from yet_another_wizz_amd import engine

n_ref, n_unk = (int(float(a)) for a in (sys.argv[1:3] if len(sys.argv) > 2 else ("3e6", "4e6")))
centers = yaw.AngularCoordinates(cs.patch_centers())
r, u = cs.sample(101, n_ref, with_z=True), cs.sample(202, n_unk, with_z=False, with_w=True)
t0 = time.perf_counter()
ref = yaw.Catalog.from_arrays(r["ra"], r["dec"], redshifts=r["z"], patch_centers=centers, degrees=False)
unk = yaw.Catalog.from_arrays(u["ra"], u["dec"], weights=u["w"], patch_centers=centers, degrees=False)
rmin, rmax = cs.SCALES_ARCMIN
config = yaw.Configuration.create(rmin=rmin, rmax=rmax, unit="arcmin", edges=cs.bin_edges())
ref.build_trees(config.binning.edges, closed=config.binning.closed); unk.build_trees(None)
print(f"set-up {time.perf_counter() - t0:.2f} s")
for key in ("hist_copies_log2", "band_cap", "tile_r"):
    if os.environ.get("YAW_" + key.upper()):
        engine.get_context().set_option(key, int(os.environ["YAW_" + key.upper()]))
for name, cats in (("cross", (ref, unk)), ("auto", (ref,))):
    links = yaw.PatchLinkage.from_catalogs(config, *cats)
    for kernel in os.environ.get("YAW_KERNELS", "auto band sweep").split():
        engine.default_kernel = kernel
        links.count_pairs(*cats)
        t0 = time.perf_counter()
        for _ in range(3):
            res = links.count_pairs(*cats)
        dt = (time.perf_counter() - t0) / 3
        st = links.last_stats
        found = sum(float(x.counts.counts.sum()) for x in res)
        print(f"{name} {kernel}: {dt*1e3:.2f} ms/call, kernels {st.kernel_ms:.2f} ms, count {st.count_ms:.2f} ms, used {st.kernel_used}, "
              f"mode {st.layout_mode}, candidates {st.candidate_pairs:.3e}, evaluated {st.evaluated_pairs:.3e}, found {found:.3e}")
