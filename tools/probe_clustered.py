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
for _opt in os.environ.get("YAW_SET", "").split(","):  # YAW_SET=key=value,key=value: any context option
    if "=" in _opt:
        engine.get_context().set_option(_opt.split("=")[0], int(_opt.split("=")[1]))
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

# the complete measurements of tests/test_gpu_clustered.py::test_full_measurement_* (reference: 10.8 s + 4.1 s on 8 cores)
if os.environ.get("YAW_FULL", "1") != "0":
    from make_golden_clustered_params import FULL
    engine.default_kernel = "auto"

    def cat(seed, n, frac, with_z, with_w):
        c = cs.sample(seed, int(n), clustered_fraction=frac, with_z=with_z, with_w=with_w)
        return yaw.Catalog.from_arrays(c["ra"], c["dec"], redshifts=c.get("z"), weights=c.get("w"), patch_centers=centers,
                                       degrees=False)

    t0 = time.perf_counter()
    fr, fu = cat(101, FULL["n_ref"], 0.7, True, False), cat(202, FULL["n_unk"], 0.7, False, True)
    frr, fur = cat(303, FULL["n_ref_rand"], 0.0, True, False), cat(404, FULL["n_unk_rand"], 0.0, False, True)
    fconfig = yaw.Configuration.create(rmin=FULL["rmin"], rmax=FULL["rmax"], unit=FULL["unit"], rweight=FULL["rweight"],
                                       resolution=FULL["resolution"], edges=cs.bin_edges())
    t1 = time.perf_counter()
    yaw.crosscorrelate(fconfig, fr, fu, ref_rand=frr, unk_rand=fur)
    t2 = time.perf_counter()
    yaw.crosscorrelate(fconfig, fr, fu, ref_rand=frr, unk_rand=fur)
    t3 = time.perf_counter()
    yaw.autocorrelate(fconfig, fr, frr)
    t4 = time.perf_counter()
    yaw.autocorrelate(fconfig, fr, frr)
    t5 = time.perf_counter()
    print(f"full measurement (kpc scales, rweight, weights, two random samples; 5.2 M objects): catalogues {t1 - t0:.2f} s, "
          f"crosscorrelate first call {(t2 - t1) * 1e3:.0f} ms (layouts + uploads), again {(t3 - t2) * 1e3:.1f} ms; "
          f"autocorrelate first {(t4 - t3) * 1e3:.0f} ms, again {(t5 - t4) * 1e3:.1f} ms")
