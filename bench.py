#!/usr/bin/env python3
"""bench.py -- headline benchmark of the angular pair-counting hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input: the DD pair count of a
cross-correlation (``PatchLinkage.count_pairs(reference, unknown)``, the call the reference makes at
src/yaw/correlation/measurements.py:623) over every linked patch pair and redshift bin, with both
catalogues already resident in HBM.  The workload is the configuration BASELINE.json quotes the
metric on: 10M reference x 10M unknown objects, uniform full sky, 30 linear z-bins, 64 patches,
one 1-10 arcmin annulus.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 5

Rank 0 prints ONE JSON line. ``value`` = candidate pairs (sum over linked patch pairs and bins of
N1*N2, what a brute-force count must decide) of the whole job divided by the slowest rank's time.
With N ranks the fixed job list is sharded (strong scaling) and the result tensor all-reduced.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VECTOR_PEAK_TFLOPS = 78.6            # MI355X FP64 vector peak incl. FMA (AMD spec)
FP32_VECTOR_PEAK_TFLOPS = 157.3           # MI355X_MICROARCH.md, chip-level parameters
HBM_PEAK_GBPS = 8000.0                    # MI355X_MICROARCH.md: HBM3E 8 TB/s
N_SIMDS = 1024                            # 256 CUs x 4 SIMDs
SHADER_CLOCK_HZ = 2.4e9                   # MI355X_MICROARCH.md: peak engine clock
WALK_VALU_PER_TRIP = 14                   # k_count_band32, one annulus, two objects per lane: vector instructions per trip


def parse_args():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n-ref", type=float, default=10e6)
    ap.add_argument("--n-unk", type=float, default=10e6)
    ap.add_argument("--patches", type=int, default=64)
    ap.add_argument("--zbins", type=int, default=30)
    ap.add_argument("--kernel", default="auto", choices=["auto", "exact", "filter", "sweep", "band"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--tile-r", type=int, default=0, help="objects per lane (0 = library default)")
    ap.add_argument("--debug-no-hits", action="store_true", help="diagnostics: time the pre-filter only (wrong counts)")
    ap.add_argument("--strip-micro", type=int, default=None, help="strip grid spacing in 1e-6 chord units (0 = no strips)")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE", help="library tunable (yawhip_ctx_set_option), e.g. band_grid_div=2")
    ap.add_argument("--scales", type=int, default=1, choices=[1, 3],
                    help="1: one scale 1-10 arcmin; 3: the log-spaced scales of BASELINE config #5 (0.5-15.8 arcmin)")
    ap.add_argument("--weights", action="store_true", help="per-object weights w ~ U(0.5, 1.5) on both catalogues")
    ap.add_argument("--kpc", action="store_true", help="physical scales 100-1000 kpc (thresholds differ from bin to bin) instead of 1-10 arcmin")
    ap.add_argument("--rweight", type=float, default=None, help="separation weight r**rweight (Configuration rweight; resolution 50)")
    ap.add_argument("--auto-randoms", type=float, default=0.0,
                    help="> 0: BASELINE config #4 -- a step is the three pair counts of an autocorrelation (DD, DR, RR) of --n-ref "
                         "data objects and this many randoms, both binned in redshift, submitted as yaw.autocorrelate submits them "
                         "(one batch); --n-unk is ignored; one GPU")
    ap.add_argument("--no-probe", action="store_true", help="with --gpus > 1: skip the scaling probe (BASELINE config #5) beside the headline")
    return ap.parse_args()


def traffic_key(args, kernel_name, band_variant=None):
    """Key of profiles/pmc_traffic.json: the whole workload -- sizes, scales, weights, separation weights (``rw<resolution>``),
    physical scales (``kpc``: per-bin thresholds, another kernel variant) -- and the band kernel variant that ran
    (``stats.band_variant``), so that a line can only quote counters of its own kernel."""
    key = (f"{kernel_name}:{int(args.n_ref)}x{int(args.n_unk)}:p{args.patches}:b{args.zbins}:s{getattr(args, 'scales', 1)}"
           f":w{int(bool(getattr(args, 'weights', False)))}")
    if getattr(args, "rweight", None) is not None:
        key += ":rw50"   # make_catalogs: resolution=50
    if getattr(args, "kpc", False):
        key += ":kpc"
    if band_variant:
        key += f":v{int(band_variant)}"
    return key


# ---------------------------------------------------------------------------------------------- inputs
def fibonacci_centers(num):
    """num near-uniform points on the sphere (patch centres for the full-sky configs)."""
    i = np.arange(num) + 0.5
    dec = np.arcsin(1.0 - 2.0 * i / num)
    ra = (np.pi * (1.0 + 5.0**0.5) * i) % (2.0 * np.pi)
    return np.column_stack([ra, dec])


def uniform_sky(seed, n):
    """ra ~ U(0, 2pi), dec = arcsin(U(-1, 1)): the recipe of the reference's BoxRandoms
    (src/yaw/randoms.py:246-259) on the full sky. Radian."""
    rng = np.random.default_rng(seed)
    return rng.uniform(0.0, 2.0 * np.pi, n), np.arcsin(rng.uniform(-1.0, 1.0, n)), rng


def box_sky(seed, n, width=60.0, height=30.0):
    """BASELINE config #2's footprint: ra ~ U(0, width), dec = arcsin(U(0, sin height)) -- a box of
    width x height degrees on the equator. DEGREES. The same recipe tools/time_reference.py --box feeds to the reference."""
    rng = np.random.default_rng(seed)
    ra = rng.uniform(0.0, width, n)
    dec = np.rad2deg(np.arcsin(rng.uniform(0.0, np.sin(np.deg2rad(height)), n)))
    return ra, dec, rng


def box_centers(width=60.0, height=30.0, grid=4):
    """Patch centres of the box: a regular grid x grid lattice. Radian, [grid * grid, 2]."""
    ga = np.linspace(width / (2 * grid), width * (2 * grid - 1) / (2 * grid), grid)
    gd = np.linspace(height / (2 * grid), height * (2 * grid - 1) / (2 * grid), grid)
    return np.deg2rad(np.array([(a, d) for a in ga for d in gd]))


def make_inputs(args):
    """The synthetic columns of the workload (seeds 101 / 202, uniform full sky, z ~ U(0.1, 1),
    w ~ U(0.5, 1.5)) -- input synthesis, not part of the catalogue set-up that ``setup_s`` times."""
    weighted = getattr(args, "weights", False)
    ra, dec, rng = uniform_sky(101, int(args.n_ref))
    ref = dict(ra=ra, dec=dec, z=rng.uniform(0.1, 1.0, len(ra)), w=rng.uniform(0.5, 1.5, len(ra)) if weighted else None)
    ra, dec, rng = uniform_sky(202, int(args.n_unk))
    unk = dict(ra=ra, dec=dec, w=rng.uniform(0.5, 1.5, len(ra)) if weighted else None)
    return ref, unk


def make_catalogs(args, inputs=None):
    import yet_another_wizz_amd as yaw

    centers = yaw.AngularCoordinates(fibonacci_centers(args.patches))
    cols_ref, cols_unk = inputs if inputs is not None else make_inputs(args)
    scales = getattr(args, "scales", 1)
    ref = yaw.Catalog.from_arrays(cols_ref["ra"], cols_ref["dec"], redshifts=cols_ref["z"], weights=cols_ref["w"],
                                  patch_centers=centers, degrees=False)
    unk = yaw.Catalog.from_arrays(cols_unk["ra"], cols_unk["dec"], weights=cols_unk["w"], patch_centers=centers, degrees=False)
    if scales == 3:
        rmin, rmax = [0.5, 1.58, 5.0], [1.58, 5.0, 15.8]
    else:
        rmin, rmax = 1.0, 10.0
    extra = {} if getattr(args, "rweight", None) is None else dict(rweight=args.rweight, resolution=50)
    unit = "arcmin"
    if getattr(args, "kpc", False):
        rmin, rmax, unit = 100.0, 1000.0, "kpc"
    config = yaw.Configuration.create(rmin=rmin, rmax=rmax, unit=unit, zmin=0.1, zmax=1.0, num_bins=args.zbins, **extra)
    return config, ref, unk


def make_auto_catalogs(n_data, n_rand, weighted=True, patches=64, zbins=30):
    """BASELINE config #4: data + randoms of an autocorrelation, both with redshifts (seeds 101 / 303);
    the same recipe tools/time_reference.py feeds to the reference."""
    import yet_another_wizz_amd as yaw

    centers = yaw.AngularCoordinates(fibonacci_centers(patches))

    def cat(seed, n):
        ra, dec, rng = uniform_sky(seed, int(n))
        z = rng.uniform(0.1, 1.0, len(ra))
        w = rng.uniform(0.5, 1.5, len(ra)) if weighted else None
        return yaw.Catalog.from_arrays(ra, dec, redshifts=z, weights=w, patch_centers=centers, degrees=False)

    config = yaw.Configuration.create(rmin=1.0, rmax=10.0, unit="arcmin", zmin=0.1, zmax=1.0, num_bins=zbins)
    return config, cat(101, n_data), cat(303, n_rand)


# ---------------------------------------------------------------------------------------------- cpu baseline
def cpu_baseline(links, ref, unk, budget_s):
    """Time the CPU oracle (oracle/paircount_oracle.c, brute force, OpenMP over the host's cores) on a
    bounded sample of the same workload, and check the GPU result on that sample against it."""
    from oracle import oracle
    from yet_another_wizz_amd import engine
    from yet_another_wizz_amd.measurements import angular_plans, threshold_table

    oracle.build()
    l1, l2 = ref._active_layout, unk._active_layout
    jobs = links.get_patch_pairs(ref, unk)
    t = threshold_table(angular_plans(links.config))

    def as_cat(layout):
        return dict(x=layout.x, y=layout.y, z=layout.z, w=layout.w, nb=layout.num_bins, off=layout.offsets)

    c1, c2 = as_cat(l1), as_cat(l2)
    sizes1, sizes2 = l1.segment_sizes(), l2.segment_sizes()
    if l1.num_bins > 1 and l2.num_bins > 1:  # binned x binned: only same-bin pairs are candidates
        cost = (sizes1[jobs[:, 0]].astype(np.float64) * sizes2[jobs[:, 1]]).sum(axis=1)
    else:
        cost = sizes1[jobs[:, 0]].sum(axis=1).astype(np.float64) * sizes2[jobs[:, 1]].sum(axis=1)
    # calibrate on the first two jobs, then size the sample to the budget
    t0 = time.perf_counter()
    oracle.count_jobs(c1, c2, jobs[:2], t)
    calib = time.perf_counter() - t0
    rate = max(float(cost[:2].sum()) / max(calib, 1e-6), 1.0)
    n_sample = int(np.clip(np.searchsorted(np.cumsum(cost), rate * budget_s), 1, len(jobs)))
    sample = jobs[:n_sample]
    t0 = time.perf_counter()
    exp_counts, exp_sums = oracle.count_jobs(c1, c2, sample, t)
    secs = time.perf_counter() - t0
    got, _ = engine.count_fine(l1, l2, sample, t)
    if l1.w is None and l2.w is None:  # integer counts: bit-identical
        parity = bool(np.array_equal(got, exp_counts.astype(np.float64)))
    else:                              # weighted sums: 1e-10 relative (north_star)
        parity = bool(np.allclose(got, exp_sums, rtol=1e-10, atol=0.0))
    pairs = float(cost[:n_sample].sum())
    return dict(
        value=pairs / secs, unit="pairs/s", cores=oracle.num_threads(), kind="port",
        sample=f"first {n_sample} of {len(jobs)} linked patch pairs, all {t.shape[0]} z-bins: "
               f"{pairs:.3e} candidate pairs in {secs:.1f} s (brute-force C oracle, OpenMP)",
        parity_with_gpu=parity,
    )


def reference_timing(args):
    """The reference's OWN count_pairs on this workload, timed in the build container by tools/time_reference.py (the
    reference cannot travel to the GPU box): seconds and effective candidate pairs/s with its multiprocessing pool on all
    cores and on one core. None when no run of this configuration is committed."""
    if getattr(args, "kpc", False) or getattr(args, "rweight", None) is not None:
        return None  # the reference was timed on angular scales without separation weights only
    name = {(10e6, 10e6, 64, 30, 1, False): "reference_cpu_10Mx10M.json",
            (10e6, 10e6, 64, 30, 1, True): "reference_cpu_10Mx10M_weighted.json",
            (1e6, 1e6, 16, 30, 1, False): None,  # config #2 was timed on its 60 x 30 degree box (reference_cpu_1Mx1M_box.json), not on the full sky
            (50e6, 50e6, 128, 30, 3, False): "reference_cpu_50Mx50M_3scales.json"}.get(
        (float(args.n_ref), float(args.n_unk), args.patches, args.zbins, getattr(args, "scales", 1),
         bool(getattr(args, "weights", False))))
    path = os.path.join(ROOT, "profiles", name) if name else None
    if not path or not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    runs = [r for r in d["runs"] if r.get("count", "DD") == "DD"]
    best = min(runs, key=lambda r: r["seconds"])
    out = dict(seconds=best["seconds"], effective_pairs_per_s=d["candidate_pairs"] / best["seconds"], cores=best["workers"],
               cpu_model=d["cpu_model"], scipy=d["scipy"], where=d["where"], source=f"profiles/{name}",
               what=d["what"], found_pairs=d["found_pairs"], tree_build_s=d["tree_build_s"])
    single = [r for r in runs if r["workers"] == 1]
    if single:
        out["one_core_seconds"] = single[0]["seconds"]
    return out


def exact_sample(links, ref, unk, n_jobs=2):
    """The brute-force FP64 kernel (the algorithm the north star describes: every candidate pair, 8 non-FMA FP64 flop)
    on a few whole jobs of the same workload, inside the same run: its FP64-VALU roofline fraction, and one more parity
    check of the default path against it."""
    from yet_another_wizz_amd import engine
    from yet_another_wizz_amd.measurements import angular_plans, threshold_table

    l1, l2 = ref._active_layout, unk._active_layout
    jobs = links.get_patch_pairs(ref, unk)[:n_jobs]
    t = threshold_table(angular_plans(links.config))
    engine.count_fine(l1, l2, jobs, t, kernel="exact")  # warm
    f_exact, st = engine.count_fine(l1, l2, jobs, t, kernel="exact")
    f_default, _ = engine.count_fine(l1, l2, jobs, t)
    k_s = max(st.count_ms, 1e-9) / 1e3
    tf = st.candidate_pairs * 8.0 / k_s / 1e12
    return dict(bound="valu_fp64", kernel="k_count (exact path)", jobs=int(len(jobs)), candidate_pairs=int(st.candidate_pairs),
                launch_ms=st.count_ms, achieved=tf, peak=FP64_VECTOR_PEAK_TFLOPS / 2.0, unit="TFLOP/s",
                frac=tf / (FP64_VECTOR_PEAK_TFLOPS / 2.0), pairs_per_s=st.candidate_pairs / k_s,
                parity_with_default_path=bool(np.array_equal(f_exact, f_default)),
                note="8 non-FMA FP64 flop per candidate pair against half the FP64 vector peak (SURVEY.md 8(d))")


def parallel_device():
    from yet_another_wizz_amd import engine

    return engine.get_context().device


def scaling_probe(steps, warmup, barrier, dist, world, rank):
    """With N > 1 ranks the headline call (0.35 ms of count kernel, ~0.2 ms of fixed cost per call) is too short to show
    what N GPUs buy; the same run therefore also times BASELINE config #5 (50M x 50M, 128 patches, 3 log scales: ~18 ms of
    count kernel on one GPU), sharded and reduced exactly like the headline. Returns the record rank 0 prints as
    ``scaling_probe`` inside the one JSON line."""
    import types

    import torch

    from yet_another_wizz_amd import PatchLinkage

    args5 = types.SimpleNamespace(n_ref=50e6, n_unk=50e6, patches=128, zbins=30, scales=3, weights=False)
    t0 = time.perf_counter()
    config, ref, unk = make_catalogs(args5)
    ref.build_trees(config.binning.edges, closed=config.binning.closed)
    unk.build_trees(None)
    links = PatchLinkage.from_catalogs(config, ref, unk)
    setup_s = time.perf_counter() - t0
    first_info = None
    for _ in range(max(warmup, 1)):
        links.count_pairs(ref, unk)
        first_info = first_info or links.last_rank_info
    barrier()
    t0 = time.perf_counter()
    count_ms_sum, reduce_ms, copy_ms = 0.0, 0.0, 0.0
    for _ in range(steps):
        links.count_pairs(ref, unk)
        count_ms_sum += links.last_stats.count_ms
        if links.last_rank_info:
            reduce_ms += links.last_rank_info["allreduce_ms"]
            copy_ms += links.last_rank_info["copy_back_ms"]
    barrier()
    elapsed = time.perf_counter() - t0
    stats = links.last_stats
    multi = multi_gpu_record(dist, world, rank, parallel_device(), links, first_info, count_ms_sum, reduce_ms, copy_ms, steps)
    stat_dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    tens = torch.tensor([elapsed], dtype=torch.float64, device=stat_dev)
    work = torch.tensor([float(stats.candidate_pairs), float(stats.count_ms)], dtype=torch.float64, device=stat_dev)
    dist.all_reduce(tens, op=dist.ReduceOp.MAX)
    cand = work.clone()
    dist.all_reduce(cand, op=dist.ReduceOp.SUM)
    dist.all_reduce(work, op=dist.ReduceOp.MAX)
    elapsed = float(tens[0])
    ref.drop_layouts()
    unk.drop_layouts()
    return dict(workload="50000000 ref x 50000000 unk uniform full sky, 30 z-bins, 128 patches, 3 log scales 0.5-15.8 arcmin, "
                         "DD count of crosscorrelate (BASELINE config #5)",
                metric="candidate pairs/s", value=float(cand[0]) * steps / elapsed, unit="pairs/s", n_gpus=world, steps=steps,
                ms_per_step=elapsed / steps * 1e3, slowest_rank_count_kernel_ms=float(work[1]), setup_s=setup_s,
                scaling="strong", multi_gpu=multi)


def multi_gpu_record(dist, world, rank, device, links, first_info, count_ms_sum, reduce_ms, copy_ms, steps):
    """What lets a reader VERIFY a multi-GPU line (gathered from every rank, printed by rank 0 as ``multi_gpu``): the
    process-group backend, the world size torch.distributed reports, the device every rank counted and reduced on, its
    share of the job list, its mean count-kernel / all-reduce / copy-back time per timed step, and what the first call paid
    once for deriving and broadcasting the job partition (``partition_ms``: outside the timed steps of this bench, inside
    every fresh ``crosscorrelate``)."""
    import torch

    info = links.last_rank_info or {}
    mine = dict(rank=rank, device=int(device), device_name=torch.cuda.get_device_name(device), jobs=info.get("jobs"),
                route=info.get("route"), count_kernel_ms=count_ms_sum / max(steps, 1), allreduce_ms=reduce_ms / max(steps, 1),
                copy_back_ms=copy_ms / max(steps, 1), partition_ms=(first_info or {}).get("partition_ms"),
                pid=os.getpid(), local_rank=int(os.environ.get("LOCAL_RANK", "0")))
    box = [None] * world
    dist.all_gather_object(box, mine)
    if rank != 0:
        return None
    return dict(backend=dist.get_backend(), world_size=dist.get_world_size(), ranks=box,
                distinct_devices=len({(r["device"]) for r in box}),
                note="one process per GPU: every rank counts its share of the linked patch pairs (LPT over the item builder's "
                     "work estimate), its rows stay in HBM, ONE sum all-reduce of the [jobs, B, E-1] tensor (RCCL when backend "
                     "= nccl) completes the result on every rank")


def main_autocorrelation(args):
    """BASELINE config #4 (--auto-randoms): one step = DD + DR + RR of an autocorrelation, submitted as ``yaw.autocorrelate``
    submits them (``PatchLinkage.count_pairs_batch``: one library call, three counts on the stream). One GPU."""
    import gc

    import torch

    from yet_another_wizz_amd import PatchLinkage, engine
    from yet_another_wizz_amd.build import source_sha16

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the pair-count path has no CPU fallback")
    for item in args.set:
        key, value = item.split("=", 1)
        engine.get_context().set_option(key, int(value))
    t_setup = time.perf_counter()
    config, data, rand = make_auto_catalogs(args.n_ref, args.auto_randoms, weighted=args.weights, patches=args.patches, zbins=args.zbins)
    data.build_trees(config.binning.edges, closed=config.binning.closed)
    rand.build_trees(config.binning.edges, closed=config.binning.closed)
    links = PatchLinkage.from_catalogs(config, data, rand)
    setup_s = time.perf_counter() - t_setup
    requests = [((data,), "DD"), ((data, rand), "DR"), ((rand,), "RR")]
    t_up = time.perf_counter()
    links.count_pairs_batch(requests)  # uploads, layouts, plans
    upload_s = time.perf_counter() - t_up
    gc.collect()
    gc.freeze()
    for _ in range(max(args.warmup, 0)):
        links.count_pairs_batch(requests)
    torch.cuda.synchronize()
    per = {k: dict(count_ms=0.0, kernel_ms=0.0) for k in ("DD", "DR", "RR")}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        links.count_pairs_batch(requests)
        for k, st in links.last_batch_stats.items():
            per[k]["count_ms"] += st.count_ms
            per[k]["kernel_ms"] += st.kernel_ms
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    steps = max(args.steps, 1)
    stats = links.last_batch_stats
    cand = float(sum(st.candidate_pairs for st in stats.values()))
    evaluated = float(sum(st.evaluated_pairs for st in stats.values()))
    peak_nofma = FP64_VECTOR_PEAK_TFLOPS / 2.0
    counts = {}
    traffic_table = {}
    pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_file):
        with open(pmc_file) as f:
            traffic_table = json.load(f)
    for k, st in stats.items():
        ms = per[k]["count_ms"] / steps
        k_s = max(ms, 1e-9) / 1e3
        tkey = f"autocorr:{k}:{int(args.n_ref)}+{int(args.auto_randoms)}:p{args.patches}:b{args.zbins}:w{int(bool(args.weights))}:v{st.band_variant}"
        entry = traffic_table.get(tkey) if isinstance(traffic_table.get(tkey), dict) else None
        fresh = bool(entry) and entry.get("source_sha16") == source_sha16()
        traffic = entry.get("bytes") if fresh else None
        sq_active = entry.get("sq_active_inst_valu") if fresh else None
        counts[k] = dict(
            candidate_pairs=int(st.candidate_pairs), evaluated_entries=int(st.evaluated_pairs), work_items=int(st.n_workgroups),
            exact_reevaluations=int(st.exact_reevaluations), count_kernel_ms=ms, all_kernels_ms=per[k]["kernel_ms"] / steps,
            band_variant=int(st.band_variant), layout_mode=int(st.layout_mode), merged_triples=int(st.merged_triples),
            fp64_equiv_frac=st.evaluated_pairs * 8.0 / k_s / 1e12 / peak_nofma,
            traffic=traffic, achieved_hbm_gbps=(traffic / k_s / 1e9 if traffic else None),
            valu_issue_frac=(sq_active * 4.0 / (N_SIMDS * SHADER_CLOCK_HZ * k_s) if sq_active else None),
            traffic_key=tkey, traffic_matches_current_sources=fresh if entry else None)
    dom = max(counts, key=lambda k: counts[k]["count_kernel_ms"])
    d = counts[dom]
    ev_tflops = d["evaluated_entries"] * 8.0 / (d["count_kernel_ms"] / 1e3) / 1e12
    roofline = dict(
        bound="valu_fp64", kernel=f"k_count_band32_one ({dom} count: the dominant launch of the step)", achieved=ev_tflops, peak=peak_nofma,
        unit="TFLOP/s", frac=ev_tflops / peak_nofma, launch_ms=d["count_kernel_ms"], traffic=d["traffic"],
        achieved_hbm_gbps=d["achieved_hbm_gbps"], achieved_hbm_frac=(d["achieved_hbm_gbps"] / HBM_PEAK_GBPS if d["achieved_hbm_gbps"] else None),
        valu_issue_frac=d["valu_issue_frac"], counts=counts,
        count_kernels_ms=sum(c["count_kernel_ms"] for c in counts.values()),
        fixed_cost_ms=elapsed / steps * 1e3 - sum(c["count_kernel_ms"] for c in counts.values()),
        note="per-bin items of binned x binned counts: a few hundred evaluations per work item -- the time is per-item latency "
             "under five to six waves per SIMD (SQ passes: profiles/r04_autocorr_10M_100M_*_sq_counters.json), not arithmetic; "
             "frac = evaluated entries x 8 FP64-equivalent flop / count-kernel time over the FP64 vector peak without FMA")
    ref_t = None
    name = "reference_cpu_auto_10M_100M_weighted.json"
    if (float(args.n_ref), float(args.auto_randoms), args.patches, args.zbins, bool(args.weights)) == (10e6, 100e6, 64, 30, True) \
            and os.path.exists(os.path.join(ROOT, "profiles", name)):
        with open(os.path.join(ROOT, "profiles", name)) as f:
            dref = json.load(f)
        best = {}
        for r in dref["runs"]:
            best[r["count"]] = min(best.get(r["count"], 1e99), r["seconds"])
        ref_t = dict(seconds=sum(best.values()), per_count_seconds=best, cores=max(r["workers"] for r in dref["runs"]),
                     cpu_model=dref["cpu_model"], scipy=dref["scipy"], where=dref["where"], source=f"profiles/{name}", what=dref["what"])
    base = None
    if args.cpu_seconds > 0:
        base = cpu_baseline(links, data, rand, args.cpu_seconds)  # the DR count's first jobs: binned x binned brute force
        base["reference"] = ref_t
    line = dict(
        metric="candidate pairs/s", value=cand * steps / elapsed, unit="pairs/s", n_gpus=1, steps=args.steps, warmup=args.warmup,
        ms_per_step=elapsed / steps * 1e3, higher_is_better=True, scaling="strong", vs_baseline=None, dtype="f64", data="synthetic",
        config=dict(workload=f"{int(args.n_ref)} data + {int(args.auto_randoms)} randoms uniform full sky, {args.zbins} z-bins, "
                             f"{args.patches} patches, 1 scale 1-10 arcmin" + (", weighted" if args.weights else "")
                             + ", DD + DR + RR of autocorrelate (BASELINE config #4)",
                    n_data=int(args.n_ref), n_random=int(args.auto_randoms), z_bins=args.zbins, patches=args.patches,
                    parallelism="one GPU, three counts in one submission"),
        candidate_pairs_per_step=cand, evaluated_pairs_per_step=evaluated, setup_s=setup_s, upload_s=upload_s, roofline=roofline,
        cpu_baseline=base)
    print(json.dumps(line), flush=True)


# ---------------------------------------------------------------------------------------------- main
def main():
    args = parse_args()
    if args.auto_randoms > 0:
        if int(os.environ.get("WORLD_SIZE", "1")) != 1 or args.gpus != 1:
            raise SystemExit("--auto-randoms runs on one GPU")
        return main_autocorrelation(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the pair-count path has no CPU fallback")
    from yet_another_wizz_amd import parallel

    device = parallel.local_device_index()  # LOCAL_RANK (YAW_AMD_DEVICE lets a rehearsal share one GPU)
    torch.cuda.set_device(device)
    if world > 1:
        # RCCL (backend "nccl") in production; YAW_BENCH_BACKEND=gloo lets several ranks rehearse on ONE GPU
        backend = os.environ.get("YAW_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)

    from yet_another_wizz_amd import PatchLinkage, engine

    engine.default_kernel = args.kernel
    if args.tile_r:
        engine.get_context().set_option("tile_r", args.tile_r)
    if args.debug_no_hits:
        engine.get_context().set_option("debug_no_hits", 1)
    if args.strip_micro is not None:
        engine.forced_strip_micro = args.strip_micro
    for item in args.set:
        key, value = item.split("=", 1)
        engine.get_context().set_option(key, int(value))
    inputs = make_inputs(args)
    t_setup = time.perf_counter()  # catalogue set-up: unit vectors, patch assignment, (patch, bin) layouts, linkage
    config, ref, unk = make_catalogs(args, inputs)
    del inputs
    ref.build_trees(config.binning.edges, closed=config.binning.closed)
    unk.build_trees(None)
    links = PatchLinkage.from_catalogs(config, ref, unk)
    setup_s = time.perf_counter() - t_setup

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        links.count_pairs(ref, unk)
        return links.last_stats

    # inputs must be resident in HBM before the timed region: upload (PCIe + device-side ordering) now
    t_up = time.perf_counter()
    micro = engine.forced_strip_micro
    if micro is None:
        micro = engine.strip_micro_for(links._angular_setup()[1])  # the spacing count_pairs will ask for
    engine.device_catalog(ref._active_layout, sort_axis=links.sort_axis, strip_micro=micro, exact=True)
    engine.device_catalog(unk._active_layout, sort_axis=links.sort_axis, strip_micro=micro, exact=True)
    # ... and every layout the count will read is built now (orientations, float32 images, merged triple runs: built on first use
    # by the library, ~20 ms per 10 M objects) -- one pass of the item builder, nothing is counted -- so that even a run with
    # --warmup 0 times counting, not data preparation
    engine.job_work(ref._active_layout, unk._active_layout, links.get_patch_pairs(ref, unk), links._angular_setup()[1],
                    sort_axis=links.sort_axis)
    upload_s = time.perf_counter() - t_up
    # Housekeeping of the interpreter, not of the path: a generation-2 garbage collection walks every object of the imported
    # modules (torch brings about a million) and takes ~40 ms -- when the allocation counters happen to trigger one inside a
    # 3 - 11 ms timed region the step time reads 5 x too long (seen on the 1M x 1M configuration). Collect now, and keep what
    # exists out of later collections.
    import gc

    gc.collect()
    gc.freeze()
    barrier()
    first_info = None  # the first call of a sharded count derives and broadcasts the job partition (partition_ms): keep its record
    for _ in range(max(args.warmup, 0)):
        step()
        first_info = first_info or links.last_rank_info
    barrier()
    t0 = time.perf_counter()
    kernel_ms, count_ms_sum, stats = 0.0, 0.0, None
    reduce_ms, copy_ms = 0.0, 0.0
    for _ in range(args.steps):
        stats = step()
        kernel_ms += stats.kernel_ms          # HIP events on the library's stream: first to last kernel of the call
        count_ms_sum += stats.count_ms        # ... and around the count kernel alone
        first_info = first_info or links.last_rank_info
        if links.last_rank_info:
            reduce_ms += links.last_rank_info["allreduce_ms"]
            copy_ms += links.last_rank_info["copy_back_ms"]
    barrier()
    elapsed = time.perf_counter() - t0
    multi = multi_gpu_record(dist, world, rank, device, links, first_info, count_ms_sum, reduce_ms, copy_ms, args.steps) if world > 1 else None

    # whole-job numbers: max time over ranks, sum of per-rank work
    stat_dev = "cuda" if (world == 1 or dist.get_backend() == "nccl") else "cpu"
    tens = torch.tensor([elapsed, kernel_ms / max(args.steps, 1)], dtype=torch.float64, device=stat_dev)
    work = torch.tensor([float(stats.candidate_pairs), float(stats.evaluated_pairs), float(stats.algorithmic_bytes),
                         float(stats.n_workgroups)], dtype=torch.float64, device=stat_dev)
    if world > 1:
        dist.all_reduce(tens, op=dist.ReduceOp.MAX)
        dist.all_reduce(work, op=dist.ReduceOp.SUM)
    elapsed, kernel_ms_step = tens.tolist()
    cand, evaluated, abytes, n_wg = work.tolist()

    if rank == 0:
        value = cand * args.steps / elapsed
        # Dominant kernel = the count kernel; its duration = the mean over rank 0's timed launches.
        #   exact : FP64 brute force, SURVEY.md 8(d): 8 non-FMA FP64 flop per candidate pair against half of the
        #           FP64 vector peak -- the roofline of the algorithm the north star describes;
        #   filter/sweep: the culling kernels evaluate only a fraction of the candidates, so the brute-force flop
        #           model no longer bounds them. What every job must still do is read both patches once,
        #           SURVEY.md 8(d)'s algorithmic bytes Bobj*(N1+N2) per job -> HBM roofline, as BASELINE.json's
        #           metric asks. The FP32 pre-filter rate and the brute-force-equivalent rate are reported beside it.
        count_ms = count_ms_sum / max(args.steps, 1) if count_ms_sum > 0 else kernel_ms / max(args.steps, 1)
        k_s = max(count_ms, 1e-9) / 1e3
        kernel_name = {1: "exact", 2: "filter", 3: "sweep", 4: "band"}.get(stats.kernel_used, str(stats.kernel_used))
        # HBM traffic and SQ counters of the count kernel are not measurable from inside this process (rocprofv3 --pmc
        # passes); they are quoted from the committed PMC run of the SAME configuration and kernel -- and only while the
        # kernel sources still hash to what that run was made with (build.source_sha16), else null
        from yet_another_wizz_amd.build import source_sha16

        traffic, traffic_source, sq_valu, sq_active_valu = None, None, None, None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        tkey = traffic_key(args, kernel_name, getattr(stats, "band_variant", 0) if stats.kernel_used == 4 else None)
        if os.path.exists(pmc_file):
            with open(pmc_file) as f:
                entry = json.load(f).get(tkey)
            if isinstance(entry, dict):
                fresh = entry.get("source_sha16") == source_sha16()
                traffic_source = {k: entry.get(k) for k in ("source", "sq_source", "commit", "date", "method", "source_sha16")}
                traffic_source["key"] = tkey
                traffic_source["matches_current_sources"] = fresh
                if fresh:
                    traffic = entry.get("bytes")
                    sq_valu = entry.get("sq_insts_valu")
                    sq_active_valu = entry.get("sq_active_inst_valu")
        fp64_equiv = stats.candidate_pairs * 8.0 / k_s / 1e12
        hbm_gbps = stats.algorithmic_bytes / k_s / 1e9
        fp32_tflops = stats.evaluated_pairs * 5.0 / k_s / 1e12
        peak_nofma = FP64_VECTOR_PEAK_TFLOPS / 2.0
        if stats.kernel_used == 1:
            roofline = dict(
                bound="valu_fp64", achieved=fp64_equiv, peak=peak_nofma, unit="TFLOP/s",
                frac=fp64_equiv / peak_nofma, traffic=traffic, traffic_source=traffic_source,
                note="FP64 vector ALU, no FMA allowed by the parity contract: 8 flop per candidate pair",
            )
        elif stats.kernel_used == 4:
            # BAND: what binds the kernel is vector-ALU issue, not HBM. The roofline is
            # the parity contract's predicate -- 8 non-FMA FP64 flop -- over the entries the kernel really EVALUATES, against
            # half the FP64 vector peak. (The kernel decides an entry in float32 wherever float32 can and in float64 inside the
            # guard bands, so this is an FP64-EQUIVALENT rate: it can exceed what FP64 arithmetic could reach.)
            ev_tflops = stats.evaluated_pairs * 8.0 / k_s / 1e12
            # The arithmetic the float32 band kernel EXECUTES: per evaluated entry 3 subtractions + 3 fused multiply-adds in
            # packed float32 = 9 flop -> against the FP32 vector peak; and the share of the kernel's time its SIMDs spend
            # issuing vector instructions at all (SQ_ACTIVE_INST_VALU counts quad-cycles: x 4 cycles, over 1024 SIMDs x clock x time).
            fp32_exec = stats.evaluated_pairs * 9.0 / k_s / 1e12
            is32 = getattr(stats, "band_variant", 0) in (32, 33)
            walk_insts = stats.evaluated_pairs / 128.0 * WALK_VALU_PER_TRIP  # an ideal walk: every trip 64 lanes x 2 evaluations
            roofline = dict(
                bound="valu_fp64", achieved=ev_tflops, peak=peak_nofma, unit="TFLOP/s", frac=ev_tflops / peak_nofma,
                traffic=traffic, traffic_source=traffic_source,
                achieved_hbm_gbps=(traffic / k_s / 1e9 if traffic else None),
                achieved_hbm_frac=(traffic / k_s / 1e9 / HBM_PEAK_GBPS if traffic else None),
                evaluated_entries_per_launch=stats.evaluated_pairs,
                fp32_frac=(fp32_exec / FP32_VECTOR_PEAK_TFLOPS if is32 else None),
                fp32_achieved_tflops=(fp32_exec if is32 else None), fp32_peak_tflops=FP32_VECTOR_PEAK_TFLOPS,
                valu_issue_frac=(sq_active_valu * 4.0 / (N_SIMDS * SHADER_CLOCK_HZ * k_s) if sq_active_valu else None),
                essential_valu_frac=(walk_insts / sq_valu if sq_valu and getattr(stats, "band_variant", 0) == 32 and args.scales == 1 else None),
                note="achieved = evaluated band entries x 8 FP64-equivalent flop / count-kernel time, peak = FP64 vector peak "
                     "without FMA (an FP64-EQUIVALENT rate: the kernel classifies in packed float32, so this is not the "
                     "utilisation of a hardware unit). Of the hardware it does use: fp32_frac = entries x 9 float32 flop "
                     "(3 sub + 3 fma) / time / FP32 vector peak; valu_issue_frac = SQ_ACTIVE_INST_VALU x 4 cycles / "
                     "(1024 SIMDs x 2.4 GHz x count-kernel time): the share of the kernel's time the vector ALUs are issuing; "
                     "essential_valu_frac = wave-level vector instructions an ideal walk needs (entries / 128 per trip x 14: one "
                     "annulus per bin, else null) over the SQ_INSTS_VALU counted. achieved_hbm_gbps = HBM bytes of the count kernel from the rocprofv3 PMC "
                     "run named in traffic_source ((2 * FETCH_SIZE + WRITE_SIZE) * 1024, separate passes) / count-kernel time "
                     "-- the figure BASELINE.json's metric names. Counter-based fields are null when the committed counters "
                     "were taken with other kernel sources or another kernel variant (traffic_source.key)",
                brute_force_equivalent=dict(achieved_tflops=fp64_equiv, peak_tflops=peak_nofma, frac=fp64_equiv / peak_nofma,
                                            note="candidate pairs x 8 FP64 flop / time; > 1 because culled pairs are never evaluated"),
            )
        else:
            roofline = dict(
                bound="hbm", achieved=hbm_gbps, peak=HBM_PEAK_GBPS, unit="GB/s", frac=hbm_gbps / HBM_PEAK_GBPS,
                traffic=traffic, traffic_source=traffic_source,
                note="algorithmic bytes = per linked patch pair, every object of both patches once (24 B, 32 B "
                     "weighted); traffic = HBM bytes per launch from the rocprofv3 PMC run named in traffic_source "
                     "((2 * FETCH_SIZE + WRITE_SIZE) * 1024, separate passes), null if none is committed for this "
                     "configuration and these kernel sources",
                evaluated=dict(evaluated_pairs_per_launch=stats.evaluated_pairs, flop_per_pair=5,
                               achieved_tflops=fp32_tflops, peak_tflops=FP32_VECTOR_PEAK_TFLOPS,
                               frac=fp32_tflops / FP32_VECTOR_PEAK_TFLOPS, note="FP32 pre-filter of the sweep / filter paths"),
                brute_force_equivalent=dict(achieved_tflops=fp64_equiv, peak_tflops=peak_nofma,
                                            frac=fp64_equiv / peak_nofma,
                                            note="candidate pairs x 8 FP64 flop / time; > 1 because culled pairs "
                                                 "are never evaluated"),
            )
        roofline.update(
            kernel={1: "k_count (exact path)",
                    4: {32: "k_count_band32 (band path, float32 classes + exact float64 guard bands)",
                        33: "k_count_band32_fine (band path, fine radial grid)"}.get(getattr(stats, "band_variant", 0),
                                                                                    "k_count_band (band path, float64)"),
                    }.get(stats.kernel_used, f"k_count_merged ({kernel_name} path)"),
            launch_ms=count_ms, all_kernels_ms=stats.kernel_ms,
            fixed_cost_ms=elapsed / max(args.steps, 1) * 1e3 - count_ms,  # everything of a step that is not the count kernel
            culled_fraction=1.0 - stats.evaluated_pairs / max(stats.candidate_pairs, 1),
            hbm_algorithmic_gbps=hbm_gbps, hbm_peak_gbps=HBM_PEAK_GBPS, hbm_algorithmic_frac=hbm_gbps / HBM_PEAK_GBPS,
            hbm_algorithmic_note="SECONDARY figure: SURVEY.md 8(d)'s algorithmic bytes (every object of both patches once per "
                                 "linked patch pair) / count-kernel time. Re-reads of a patch are served by L2 / MALL, so this "
                                 "is not HBM utilisation (achieved_hbm_gbps is)",
        )
        base = None
        if world == 1 and args.cpu_seconds > 0:
            base = cpu_baseline(links, ref, unk, args.cpu_seconds)
            base["reference"] = reference_timing(args)
            if stats.kernel_used != 1:
                roofline["exact_sample"] = exact_sample(links, ref, unk)
        line = dict(
            metric="candidate pairs/s", value=value, unit="pairs/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
            ms_per_step=elapsed / max(args.steps, 1) * 1e3, higher_is_better=True, scaling="strong", vs_baseline=None,
            dtype="f64", data="synthetic",
            dtype_note="every pair is decided as the float64 predicate decides it (results identical to the all-float64 kernels); "
                       "the default band kernel classifies in float32 and re-evaluates the pairs inside its guard bands in float64",
            config=dict(
                workload=f"{int(args.n_ref)} ref x {int(args.n_unk)} unk uniform full sky, {args.zbins} z-bins, "
                         f"{args.patches} patches, "
                         + ("1 scale 1-10 arcmin" if args.scales == 1 else "3 log scales 0.5-15.8 arcmin")
                         + (", weighted" if args.weights else "") + ", DD count of crosscorrelate",
                n_ref=int(args.n_ref), n_unk=int(args.n_unk), z_bins=args.zbins, patches=args.patches,
                linked_patch_pairs=int(len(links.get_patch_pairs(ref, unk))), kernel=kernel_name,
                parallelism=f"patch-pair sharding x{world}",
            ),
            candidate_pairs_per_step=cand, evaluated_pairs_per_step=evaluated,
            kernel_ms_per_step=kernel_ms_step, setup_s=setup_s, upload_s=upload_s, roofline=roofline, cpu_baseline=base,
        )
        if multi is not None:
            line["multi_gpu"] = multi
    probe = None
    if world > 1 and not args.no_probe:
        ref.drop_layouts()
        unk.drop_layouts()
        probe = scaling_probe(min(max(args.steps, 1), 5), 1, barrier, dist, world, rank)
    if rank == 0:
        if probe is not None:
            line["scaling_probe"] = probe
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
