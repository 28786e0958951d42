"""Throughput probe of the autocorrelation counts (both sides binned): DD-like self count and a DR-like
count against a 10x larger binned random catalogue, full sky, 64 patches, 30 bins."""
import sys, time, types
import numpy as np
sys.path.insert(0, ".")
import bench
import yet_another_wizz_amd as yaw
from yet_another_wizz_amd import engine
from yet_another_wizz_amd.measurements import angular_plans, threshold_table

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
nr = int(float(sys.argv[2])) if len(sys.argv) > 2 else 3 * n
centers = yaw.AngularCoordinates(bench.fibonacci_centers(64))
def cat(seed, m):
    ra, dec, rng = bench.uniform_sky(seed, m)
    return yaw.Catalog.from_arrays(ra, dec, redshifts=rng.uniform(0.1, 1.0, m), patch_centers=centers, degrees=False)
config = yaw.Configuration.create(rmin=1.0, rmax=10.0, unit="arcmin", zmin=0.1, zmax=1.0, num_bins=30)
data, rand = cat(101, n), cat(303, nr)
ld, lr = data.build_trees(config.binning.edges), rand.build_trees(config.binning.edges)
links = yaw.PatchLinkage.from_catalogs(config, data, rand)
t = threshold_table(angular_plans(config))
for name, l1, l2, auto in (("DD", ld, ld, True), ("DR", ld, lr, False), ("RR", lr, lr, True)):
    jobs = links.get_patch_pairs(data, None if auto else rand)
    for kern in ("sweep", "filter") if name == "DD" else ("sweep",):
        for rep in range(2):
            fine, st = engine.count_fine(l1, l2, jobs, t, kernel=kern)
        print(f"{name} {kern}: jobs={len(jobs)} cand={st.candidate_pairs:.3e} eval={st.evaluated_pairs:.3e} kernel_ms={st.kernel_ms:.2f} "
              f"rate={st.candidate_pairs/st.kernel_ms/1e6:.1f} Gpairs/s found={fine.sum():.4g} wgs={st.n_workgroups}", flush=True)
