"""Host-side cost of the public entry points (warm: catalogues resident): cProfile of yaw.autocorrelate and
yaw.crosscorrelate on small catalogues, where the kernels take well under a millisecond."""
import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import yet_another_wizz_amd as yaw

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
centers = yaw.AngularCoordinates(bench.fibonacci_centers(64))
def cat(seed, m, with_z):
    ra, dec, rng = bench.uniform_sky(seed, m)
    z = rng.uniform(0.1, 1.0, m) if with_z else None
    return yaw.Catalog.from_arrays(ra, dec, redshifts=z, patch_centers=centers, degrees=False)
config = yaw.Configuration.create(rmin=1.0, rmax=10.0, unit="arcmin", zmin=0.1, zmax=1.0, num_bins=30)
ref, unk, rr, ur = cat(101, n, True), cat(202, n, False), cat(303, 2 * n, True), cat(404, 2 * n, False)
for name, call in (("autocorrelate (DD, DR, RR)", lambda: yaw.autocorrelate(config, ref, rr)),
                   ("crosscorrelate (DD, DR, RD, RR)", lambda: yaw.crosscorrelate(config, ref, unk, ref_rand=rr, unk_rand=ur))):
    for _ in range(3):
        t0 = time.perf_counter(); call(); dt = time.perf_counter() - t0
    print(f"{name} warm: {dt*1e3:.2f} ms")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10):
        call()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)
