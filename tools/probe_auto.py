"""Throughput probe of the autocorrelation counts (both sides binned): DD self count, DR against a larger binned
random catalogue, RR self count of the randoms -- full sky, 64 patches, 30 bins (BASELINE config #4 with
`1e7 1e8 w`).   python tools/probe_auto.py [n_data] [n_random] [w]"""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import yet_another_wizz_amd as yaw
from yet_another_wizz_amd import engine
from yet_another_wizz_amd.measurements import angular_plans, threshold_table

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
nr = int(float(sys.argv[2])) if len(sys.argv) > 2 else 3 * n
weighted = len(sys.argv) > 3 and sys.argv[3].startswith("w")
centers = yaw.AngularCoordinates(bench.fibonacci_centers(64))
def cat(seed, m):
    ra, dec, rng = bench.uniform_sky(seed, m)
    z = rng.uniform(0.1, 1.0, m)
    w = rng.uniform(0.5, 1.5, m) if weighted else None
    return yaw.Catalog.from_arrays(ra, dec, redshifts=z, weights=w, patch_centers=centers, degrees=False)
if os.environ.get("YAW_SEG"):
    engine.get_context().set_option("seg_strips", int(os.environ["YAW_SEG"]))
if os.environ.get("YAW_SEG_MIN_RUN"):
    engine.get_context().set_option("seg_strips_min_run", int(os.environ["YAW_SEG_MIN_RUN"]))
if os.environ.get("YAW_STRIP_MICRO"):
    engine.forced_strip_micro = int(os.environ["YAW_STRIP_MICRO"])
if os.environ.get("YAW_TILE_R"):
    engine.get_context().set_option("tile_r", int(os.environ["YAW_TILE_R"]))
for _opt in os.environ.get("YAW_SET", "").split(","):  # YAW_SET=key=value,key=value: any context option
    if "=" in _opt:
        engine.get_context().set_option(_opt.split("=")[0], int(_opt.split("=")[1]))
if os.environ.get("YAW_BAND_CAP"):
    engine.get_context().set_option("band_cap", int(os.environ["YAW_BAND_CAP"]))
config = yaw.Configuration.create(rmin=1.0, rmax=10.0, unit="arcmin", zmin=0.1, zmax=1.0, num_bins=30)
t0 = time.perf_counter()
data, rand = cat(101, n), cat(303, nr)
print(f"catalogues built in {time.perf_counter() - t0:.1f} s (weighted={weighted})", flush=True)
t0 = time.perf_counter()
ld, lr = data.build_trees(config.binning.edges), rand.build_trees(config.binning.edges)
links = yaw.PatchLinkage.from_catalogs(config, data, rand)
t = threshold_table(angular_plans(config))
print(f"layouts + linkage in {time.perf_counter() - t0:.1f} s", flush=True)
total = 0.0
for name, l1, l2, auto in (("DD", ld, ld, True), ("DR", ld, lr, False), ("RR", lr, lr, True)):
    jobs = links.get_patch_pairs(data, None if auto else rand)
    for kern in ("auto", "filter") if (name == "DD" and n <= 2_000_000) else ("auto", "sweep", "band"):
        for rep in range(2):
            fine, st = engine.count_fine(l1, l2, jobs, t, kernel=kern)
        if kern == "auto":
            total += st.kernel_ms
        kern = f"{kern}->{st.kernel_used}"
        print(f"{name} {kern}: jobs={len(jobs)} cand={st.candidate_pairs:.3e} eval={st.evaluated_pairs:.3e} kernel_ms={st.kernel_ms:.2f} "
              f"count_ms={st.count_ms:.2f} mode={st.layout_mode} rate={st.candidate_pairs/st.kernel_ms/1e6:.1f} Gpairs/s found={fine.sum():.6g} wgs={st.n_workgroups}", flush=True)
print(f"DD+DR+RR kernels: {total:.2f} ms", flush=True)
# the whole measurement through the public API (second call: catalogues resident, layouts cached)
for rep in range(2):
    t0 = time.perf_counter()
    (cf,) = yaw.autocorrelate(config, data, rand)
    dt = time.perf_counter() - t0
print(f"yaw.autocorrelate end to end (warm): {dt*1e3:.1f} ms; w(z) first bins {cf.sample().data[:3]}", flush=True)
