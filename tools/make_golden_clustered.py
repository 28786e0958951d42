#!/usr/bin/env python3
"""Run the REFERENCE on the clustered synthetic survey of tests/clustered_sky.py (build container only; needs
/root/reference) and keep its pair-count tensors as tests/golden/clustered_reference_counts.npz:

    cross   PatchLinkage.count_pairs(reference, unknown)   reference binned and unweighted, unknown unbinned and weighted
    auto    PatchLinkage.count_pairs(reference)            the autocorrelation count of the reference sample

(/root/reference/src/yaw/correlation/measurements.py:307-367). tests/test_gpu_clustered.py builds the same catalogues
with this package and compares the full [scale, bin, patch, patch] tensors."""
from __future__ import annotations

import argparse
import os
import shutil
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


from make_golden_clustered_params import FULL  # tests/: shared with the GPU test


def make_full(args, yaw, cs, out_path):
    """A complete measurement at mid size on the clustered survey: crosscorrelate with both random samples (DD, DR, RD,
    RR), physical scales (every z-bin its own thresholds), separation weights (31 fine bins per z-bin), weighted unknown
    sample and weighted randoms -> count tensors, patch sums and CorrFunc.sample() of the reference."""
    import pandas as pd
    from make_golden import dump_counts
    from yaw.coordinates import AngularCoordinates

    centers = AngularCoordinates(cs.patch_centers())
    shutil.rmtree(args.cache, ignore_errors=True)
    os.makedirs(args.cache)

    def cat(name, seed, n, frac, with_z, with_w):
        cols = cs.sample(seed, int(n), clustered_fraction=frac, with_z=with_z, with_w=with_w)
        return yaw.Catalog.from_dataframe(os.path.join(args.cache, name), pd.DataFrame(cols), ra_name="ra", dec_name="dec",
                                          redshift_name="z" if with_z else None, weight_name="w" if with_w else None,
                                          patch_centers=centers, degrees=False)

    ref = cat("ref", 101, FULL["n_ref"], 0.7, True, False)
    unk = cat("unk", 202, FULL["n_unk"], 0.7, False, True)
    ref_rand = cat("ref_rand", 303, FULL["n_ref_rand"], 0.0, True, False)
    unk_rand = cat("unk_rand", 404, FULL["n_unk_rand"], 0.0, False, True)
    config = yaw.Configuration.create(rmin=FULL["rmin"], rmax=FULL["rmax"], unit=FULL["unit"], rweight=FULL["rweight"],
                                      resolution=FULL["resolution"], edges=cs.bin_edges())
    # astropy is absent here: the reference's Planck15 stand-in (tools/ref_loader.py) gets its
    # D_A(z) from this package's flat-LCDM (itself pinned by the reference's estimate.dat); everything downstream --
    # scale / D_A, edges, thresholds, counts, estimator -- is the reference's own code (cosmology.py:250-259).
    from yet_another_wizz_amd.cosmology import get_default_cosmology

    ours = get_default_cosmology()
    type(config.cosmology).angular_diameter_distance = lambda self, z: ours.angular_diameter_distance(z)
    t0 = time.perf_counter()
    cfs = yaw.crosscorrelate(config, ref, unk, ref_rand=ref_rand, unk_rand=unk_rand, max_workers=args.workers)
    secs = time.perf_counter() - t0
    out = {"seconds": secs}
    dump_counts("cross", cfs, out)
    print(f"crosscorrelate (DD, DR, RD, RR): {secs:.1f} s; w(z) = {cfs[0].sample().data}", flush=True)
    t0 = time.perf_counter()
    acfs = yaw.autocorrelate(config, ref, ref_rand, max_workers=args.workers)  # binned x binned: DD, DR, RR
    out["auto_seconds"] = time.perf_counter() - t0
    dump_counts("auto", acfs, out)
    print(f"autocorrelate (DD, DR, RR): {out['auto_seconds']:.1f} s; w(z) = {acfs[0].sample().data}", flush=True)
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, os.path.getsize(out_path), "bytes")
    shutil.rmtree(args.cache, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-ref", type=float, default=1.5e6)
    ap.add_argument("--n-unk", type=float, default=2.0e6)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--exact-slots", type=int, default=12, help="largest slots of the cross count to recompute exactly")
    ap.add_argument("--cache", default="/dev/shm/yaw_ref_clustered")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "clustered_reference_counts.npz"))
    ap.add_argument("--full", action="store_true", help="the complete mid-size measurement instead (clustered_reference_full.npz)")
    args = ap.parse_args()
    os.environ["YAW_NUM_THREADS"] = str(args.workers)
    from ref_loader import load_reference

    yaw = load_reference(args.workers)
    if args.full:
        import clustered_sky

        return make_full(args, yaw, clustered_sky, os.path.join(ROOT, "tests", "golden", "clustered_reference_full.npz"))
    import pandas as pd
    import scipy
    from yaw.coordinates import AngularCoordinates
    from yaw.correlation.measurements import PatchLinkage

    import clustered_sky as cs

    n_ref, n_unk = int(args.n_ref), int(args.n_unk)
    centers = AngularCoordinates(cs.patch_centers())
    shutil.rmtree(args.cache, ignore_errors=True)
    os.makedirs(args.cache)
    ref_cols = cs.sample(101, n_ref, with_z=True, with_w=False)
    unk_cols = cs.sample(202, n_unk, with_z=False, with_w=True)
    ref = yaw.Catalog.from_dataframe(os.path.join(args.cache, "ref"), pd.DataFrame(ref_cols), ra_name="ra", dec_name="dec",
                                     redshift_name="z", patch_centers=centers, degrees=False)
    unk = yaw.Catalog.from_dataframe(os.path.join(args.cache, "unk"), pd.DataFrame(unk_cols), ra_name="ra", dec_name="dec",
                                     weight_name="w", patch_centers=centers, degrees=False)
    rmin, rmax = cs.SCALES_ARCMIN
    config = yaw.Configuration.create(rmin=rmin, rmax=rmax, unit="arcmin", edges=cs.bin_edges())
    ref.build_trees(config.binning.edges, closed=config.binning.closed)
    unk.build_trees(None)
    links = PatchLinkage.from_catalogs(config, ref, unk)
    out = dict(n_ref=n_ref, n_unk=n_unk, scipy=scipy.__version__, numpy=np.__version__,
               num_records_ref=np.array(ref.get_num_records()), num_records_unk=np.array(unk.get_num_records()))
    for name, cats in (("cross", (ref, unk)), ("auto", (ref,))):
        t0 = time.perf_counter()
        res = links.count_pairs(*cats, max_workers=args.workers)
        secs = time.perf_counter() - t0
        counts = np.stack([r.counts.counts for r in res])  # [S, B, P, P]
        out[f"{name}_counts"] = counts
        out[f"{name}_sum_weights1"] = res[0].sum_weights.sum_weights1
        out[f"{name}_sum_weights2"] = res[0].sum_weights.sum_weights2
        out[f"{name}_seconds"] = secs
        print(f"{name}: {secs:.1f} s, {counts.sum():.6e} pairs (weighted sum), per scale {counts.sum(axis=(1, 2, 3))}", flush=True)
    # The reference's weighted sums are not exactly rounded: scipy's traversal adds products of node weight sums to
    # running totals of ~1e9. For the largest slots keep the (practically) exact value as well: integer pair counts per
    # unknown object (cKDTree.query_ball_point(..., return_length=True) at both radii of the scale) times its weight,
    # summed with math.fsum -- the only rounding left is the one of w * count.
    import math

    from scipy.spatial import cKDTree

    cross = out["cross_counts"]
    top = np.argsort(cross.ravel())[::-1][: args.exact_slots]
    idx = np.array(np.unravel_index(top, cross.shape)).T  # [K, 4] = (scale, bin, patch1, patch2)
    edges = cs.bin_edges()
    arcmin = np.pi / 180.0 / 60.0
    exact = []
    for s, k, p1, p2 in idx:
        patch1, patch2 = ref[int(p1)], unk[int(p2)]
        zbin = np.digitize(patch1.redshifts, edges, right=True) - 1  # closed = right (trees.py:408-410)
        a = patch1.coords.to_3d()[zbin == k]
        b, wb = patch2.coords.to_3d(), patch2.weights
        tree = cKDTree(a)
        r_lo, r_hi = (2.0 * np.sin(np.asarray([rmin[s], rmax[s]]) * arcmin / 2.0)).tolist()
        n_hi = tree.query_ball_point(b, r_hi, return_length=True, workers=args.workers)
        n_lo = tree.query_ball_point(b, r_lo, return_length=True, workers=args.workers)
        exact.append(math.fsum((wb * (n_hi - n_lo)).tolist()))
        print(f"slot {(int(s), int(k), int(p1), int(p2))}: reference {cross[s, k, p1, p2]!r}, exact {exact[-1]!r}, "
              f"difference {cross[s, k, p1, p2] - exact[-1]:+.3e}", flush=True)
    out["cross_exact_idx"] = idx.astype(np.int64)
    out["cross_exact_val"] = np.array(exact)
    np.savez_compressed(args.out, **out)
    print("wrote", args.out, os.path.getsize(args.out), "bytes")
    shutil.rmtree(args.cache, ignore_errors=True)


if __name__ == "__main__":
    main()
