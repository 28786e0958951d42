#!/usr/bin/env python3
"""Time the REFERENCE's own CPU path on bench.py's workload (build container only; needs /root/reference).

What is timed is exactly the call the GPU path replaces: ``PatchLinkage.count_pairs(reference, unknown)``
(/root/reference/src/yaw/correlation/measurements.py:307-367 -> process_patch_pair :88-128 ->
AngularTree.count, catalog/trees.py:303-362 -> scipy KDTree.count_neighbors) with the KD-trees already
built (the tree build is reported separately), once with one worker and once with all cores
(``max_workers``; the reference's multiprocessing pool, utils/parallel.py:251-346).
Inputs are bench.py's: same seeds, same Fibonacci patch centres, same binning and scales, so the candidate
pair count and the pair counts themselves are those of the GPU bench line.

    python tools/time_reference.py --n-ref 10e6 --n-unk 10e6 --patches 64 --out profiles/reference_cpu_10Mx10M.json

The JSON it writes is a tracked, small file that bench.py quotes as ``cpu_baseline.reference``.
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import shutil
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-ref", type=float, default=10e6)
    ap.add_argument("--n-unk", type=float, default=10e6)
    ap.add_argument("--patches", type=int, default=64)
    ap.add_argument("--zbins", type=int, default=30)
    ap.add_argument("--scales", type=int, default=1, choices=[1, 3], help="3: the log-spaced scales of BASELINE config #5")
    ap.add_argument("--weights", action="store_true", help="per-object weights w ~ U(0.5, 1.5)")
    ap.add_argument("--auto-randoms", type=float, default=0.0,
                    help="> 0: BASELINE config #4 -- autocorrelation counts DD, DR, RR of --n-ref data objects and "
                         "this many randoms (both binned); --n-unk is ignored")
    ap.add_argument("--workers", type=int, nargs="+", default=[8, 1])
    ap.add_argument("--cache", default="/dev/shm/yaw_ref_timing")
    ap.add_argument("--out", default=None)
    ap.add_argument("--slots-out", default=None,
                    help="write the reference's per-slot values counts[s][:, i, j] of every --slot-every'th linked patch "
                         "pair (and the patch ids) to this .npz: the full-size parity fixtures of tests/golden/")
    ap.add_argument("--slot-every", type=int, default=1)
    ap.add_argument("--box", default=None, metavar="WxH",
                    help="BASELINE config #2's footprint: a W x H degree box on the equator with a regular grid of patch "
                         "centres (bench.box_sky / bench.box_centers; --patches must be a square number) instead of the full sky")
    args = ap.parse_args()

    os.environ["YAW_NUM_THREADS"] = str(max(args.workers))
    from ref_loader import load_reference

    yaw = load_reference(max(args.workers))
    import pandas as pd
    import scipy
    from yaw.coordinates import AngularCoordinates
    from yaw.correlation.measurements import PatchLinkage

    import bench  # input recipe only (fibonacci_centers, uniform_sky): no GPU code is touched

    n_ref, n_unk = int(args.n_ref), int(args.n_unk)
    box = tuple(float(v) for v in args.box.split("x")) if args.box else None
    if box:
        grid = int(round(args.patches ** 0.5))
        assert grid * grid == args.patches, "--box needs a square number of patches"
        centers = AngularCoordinates(bench.box_centers(box[0], box[1], grid))
    else:
        centers = AngularCoordinates(bench.fibonacci_centers(args.patches))
    shutil.rmtree(args.cache, ignore_errors=True)
    os.makedirs(args.cache)

    auto = args.auto_randoms > 0
    n_rand = int(args.auto_randoms)

    def cached(name, seed, n, with_z):
        if box:
            ra, dec, rng = bench.box_sky(seed, n, box[0], box[1])
            ra, dec = np.deg2rad(ra), np.deg2rad(dec)
        else:
            ra, dec, rng = bench.uniform_sky(seed, n)
        cols = dict(ra=ra, dec=dec)
        if with_z:
            cols["z"] = rng.uniform(0.1, 1.0, n)
        if args.weights:
            cols["w"] = rng.uniform(0.5, 1.5, n)
        return yaw.Catalog.from_dataframe(os.path.join(args.cache, name), pd.DataFrame(cols), ra_name="ra", dec_name="dec",
                                          redshift_name="z" if with_z else None, weight_name="w" if args.weights else None,
                                          patch_centers=centers, degrees=False)

    t0 = time.perf_counter()
    ref = cached("ref", 101, n_ref, True)                                             # seeds: SURVEY.md 8(d)
    unk = cached("rand", 303, n_rand, True) if auto else cached("unk", 202, n_unk, False)
    ingest_s = time.perf_counter() - t0
    print(f"catalogues cached in {ingest_s:.1f} s", flush=True)

    rmin, rmax = ([0.5, 1.58, 5.0], [1.58, 5.0, 15.8]) if args.scales == 3 else (1.0, 10.0)
    config = yaw.Configuration.create(rmin=rmin, rmax=rmax, unit="arcmin", zmin=0.1, zmax=1.0, num_bins=args.zbins)
    t0 = time.perf_counter()
    ref.build_trees(config.binning.edges, closed=config.binning.closed)
    unk.build_trees(config.binning.edges if auto else None, closed=config.binning.closed)
    trees_s = time.perf_counter() - t0
    print(f"trees built in {trees_s:.1f} s ({max(args.workers)} workers)", flush=True)

    links = PatchLinkage.from_catalogs(config, ref, unk)
    passes = [("DD", (ref,)), ("DR", (ref, unk)), ("RR", (unk,))] if auto else [("DD", (ref, unk))]
    jobs = list(links.iter_patch_id_pairs(auto=False))
    nrec_ref = np.array([ref[i].meta.num_records for i in range(args.patches)], dtype=np.float64)
    nrec_unk = np.array([unk[i].meta.num_records for i in range(args.patches)], dtype=np.float64)
    cand = float(sum(nrec_ref[i] * nrec_unk[j] for i, j in jobs))  # patch-level candidates of the cross count

    runs = []
    totals = {}
    slots = {}
    for w in args.workers:
        for name, cats in passes:
            t0 = time.perf_counter()
            res = links.count_pairs(*cats, max_workers=w)
            secs = time.perf_counter() - t0
            per_scale_bin = np.array([r.counts.counts.sum(axis=(1, 2)) for r in res])  # [S, B]
            found = float(per_scale_bin.sum())
            if name not in totals:
                totals[name] = per_scale_bin
                if args.slots_out:
                    # jobs of this count as the reference lists them (measurements.py:258-289), every n-th kept
                    ids = np.array(list(links.iter_patch_id_pairs(auto=len(cats) == 1)), dtype=np.int32)[:: args.slot_every]
                    vals = np.stack([r.counts.counts[:, ids[:, 0], ids[:, 1]] for r in res])  # [S, B, n_sel]
                    slots[name + "_ids"] = ids
                    slots[name + "_values"] = np.ascontiguousarray(np.moveaxis(vals, 2, 0))   # [n_sel, S, B]
                    slots[name + "_sum_weights1"] = res[0].sum_weights.sum_weights1
                    slots[name + "_sum_weights2"] = res[0].sum_weights.sum_weights2
            assert np.allclose(per_scale_bin, totals[name], rtol=1e-12, atol=0)
            run = dict(count=name, workers=w, seconds=secs, found_pairs_per_s=found / secs)
            if not auto:
                run["effective_pairs_per_s"] = cand / secs
            runs.append(run)
            print(f"{name} count_pairs, {w} workers: {secs:.2f} s, {found:.6e} pairs found", flush=True)
    total = float(totals["DD"].sum())
    counts = None

    out = dict(
        what="reference PatchLinkage.count_pairs(reference, unknown), trees pre-built (measurements.py:307-367)",
        workload=(f"{n_ref} data + {n_rand} randoms autocorrelation (DD, DR, RR)" if auto else f"{n_ref} ref x {n_unk} unk")
                 + (f", uniform {args.box} degree box" if args.box else ", uniform full sky") + f", {args.zbins} z-bins, {args.patches} patches, "
                 + ("1-10 arcmin" if args.scales == 1 else "3 log scales 0.5-15.8 arcmin") + (", weighted" if args.weights else ""),
        n_ref=n_ref, n_unk=n_rand if auto else n_unk, patches=args.patches, z_bins=args.zbins, scales=args.scales,
        weighted=bool(args.weights), auto=auto, linked_patch_pairs=len(jobs),
        candidate_pairs=cand, found_pairs=total, pairs_per_bin=totals["DD"].sum(axis=0).tolist(),
        pairs_per_scale_bin={k: v.tolist() for k, v in totals.items()},
        runs=runs, tree_build_s=trees_s, ingest_s=ingest_s,
        cpu_model=_cpu_model(), cores_available=os.cpu_count(), scipy=scipy.__version__, numpy=np.__version__,
        python=platform.python_version(), where="build container (not the GPU box: the reference cannot travel)",
    )
    print(json.dumps(out))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)
    if args.slots_out:
        np.savez_compressed(args.slots_out, n_ref=n_ref, n_unk=n_rand if auto else n_unk, patches=args.patches,
                            z_bins=args.zbins, scales=args.scales, weighted=bool(args.weights), slot_every=args.slot_every,
                            candidate_pairs=cand, box=np.array(box if box else (0.0, 0.0)), **slots)
    shutil.rmtree(args.cache, ignore_errors=True)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


if __name__ == "__main__":
    main()
