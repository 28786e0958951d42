import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import clustered_sky as cs
import yet_another_wizz_amd as yaw
from yet_another_wizz_amd import engine
from make_golden_clustered_params import FULL
centers = yaw.AngularCoordinates(cs.patch_centers())
def cat(seed, n, frac, with_z, with_w):
    c = cs.sample(seed, int(n), clustered_fraction=frac, with_z=with_z, with_w=with_w)
    return yaw.Catalog.from_arrays(c["ra"], c["dec"], redshifts=c.get("z"), weights=c.get("w"), patch_centers=centers, degrees=False)
fr, frr = cat(101, FULL["n_ref"], 0.7, True, False), cat(303, FULL["n_ref_rand"], 0.0, True, False)
for rw in (FULL["rweight"], None):
    config = yaw.Configuration.create(rmin=FULL["rmin"], rmax=FULL["rmax"], unit=FULL["unit"], rweight=rw, resolution=FULL["resolution"], edges=cs.bin_edges())
    fr.build_trees(config.binning.edges); frr.build_trees(config.binning.edges)
    links = yaw.PatchLinkage.from_catalogs(config, fr, frr)
    for name, cats in (("DD", (fr,)), ("DR", (fr, frr)), ("RR", (frr,))):
        for kernel in ("auto", "band", "sweep"):
            engine.default_kernel = kernel
            links.count_pairs(*cats)
            t0 = time.perf_counter(); links.count_pairs(*cats); dt = time.perf_counter() - t0
            st = links.last_stats
            print(f"rweight={rw} {name} {kernel}: {dt*1e3:.2f} ms, kernels {st.kernel_ms:.2f}, count {st.count_ms:.2f}, used {st.kernel_used}, mode {st.layout_mode}, evaluated {st.evaluated_pairs:.3e}, wgs {st.n_workgroups}")
