"""Cross- and autocorrelation drivers on the MI355X pair-count engine.

Drop-in for the nn path of ``yaw.correlation.measurements`` (src/yaw/correlation/measurements.py):
``crosscorrelate`` (:529-628), ``autocorrelate`` (:455-525), ``PatchLinkage`` (:171-392) keep their
call shapes, error behaviour and result types.  What changes is the execution model: the
reference farms one Python task per patch pair to a process pool, each task unpickles KD-trees
and loops over redshift bins (:88-128, :344-364); here the whole job list of one pair count goes
to the GPU in a single ``yawhip_count_pairs`` call against catalogues resident in HBM, the jobs
are sharded over the processes of a ``torch.distributed`` group (one per GPU) and the dense
``[B, E-1, P, P]`` tensor is combined with one all-reduce.
"""
from __future__ import annotations

import logging
from itertools import compress

import numpy as np

from . import _lib, engine, parallel
from .angular_bins import plan_for_limits
from .catalog import Catalog, InconsistentPatchesError
from .coordinates import AngularDistances
from .corrfunc import CorrFunc
from .paircounts import NormalisedCounts, PatchedCounts, PatchedSumWeights

__all__ = ["autocorrelate", "crosscorrelate", "PatchLinkage", "get_max_angle", "check_patch_conistency"]

logger = logging.getLogger("yet_another_wizz_amd")


def _log_info(*args) -> None:
    if parallel.world()[0] == 0:
        logger.info(*args)


def check_patch_conistency(catalog: Catalog, *catalogs: Catalog, rtol: float = 0.5) -> None:
    """Patch centres of the other catalogues must lie within ``rtol`` patch radii of
    ``catalog``'s centres (measurements.py:131-149; the reference's spelling is kept)."""
    centers, radii = catalog.get_centers(), catalog.get_radii()
    for other in catalogs:
        offset = centers.distance(other.get_centers())
        if np.any(offset.data / radii.data > rtol):
            raise InconsistentPatchesError("patch centers are not aligned")


def get_max_angle(config, redshift_limit: float = 0.05) -> AngularDistances:
    """Largest separation any pair can contribute at: the upper scale limits evaluated at
    max(zmin, redshift_limit) (measurements.py:152-168)."""
    z = max(config.binning.zmin, redshift_limit)
    _, ang_max = config.scales.scales.get_angle_radian(z, cosmology=config.cosmology)
    return AngularDistances(np.max(ang_max))


def angular_plans(config):
    """One :class:`AngularBinPlan` per redshift bin: scale limits at the bin centre
    (measurements.py:99,110-112) -> edges -> thresholds."""
    scales, cosmology = config.scales.scales, config.cosmology
    return [
        plan_for_limits(*scales.get_angle_radian(zmid, cosmology=cosmology), config.scales.rweight,
                        config.scales.resolution)
        for zmid in config.binning.binning.mids
    ]


class CombinePlan:
    """``AngularBinPlan.combine`` for all redshift bins of one configuration, prepared once: bins whose plans share
    the edge count, the per-scale slices and the separation weights (always the case for angular units) are handled
    in one vectorised pass over fine[B, E-1, J] -> [S, B, J]."""

    __slots__ = ("plans", "num_scales", "uniform", "num_fine", "slices", "scale_factors", "single")

    def __init__(self, plans):
        first = plans[0]
        self.plans = plans
        self.num_scales = first.num_scales
        self.uniform = all(
            p.num_edges == first.num_edges and p._slices == first._slices
            and (p._scale_factor is None) == (first._scale_factor is None)
            for p in plans
        )
        self.num_fine = first.num_edges - 1
        self.slices = list(first._slices)
        self.scale_factors = None
        if self.uniform and first._scale_factor is not None:
            self.scale_factors = np.stack([p._scale_factor for p in plans])[:, :, np.newaxis]
        # the common case: one scale made of the one fine bin the device counted
        self.single = self.uniform and self.scale_factors is None and self.slices == [(0, 1)] and self.num_fine == 1

    def dense_spec(self, num_fine: int):
        """The recombination as ``yawhip_count_pairs_dense`` takes it: ``slices`` int32[B, S, 2] (fine bins [first, last)
        of every scale and bin) and ``factors`` f64[B, num_fine] (separation weights, 0 for the padded fine bins of a bin
        with fewer edges) or None."""
        slices = np.ascontiguousarray([p._slices for p in self.plans], dtype=np.int32)
        factors = None
        if any(p._scale_factor is not None for p in self.plans):
            factors = np.zeros((len(self.plans), num_fine), dtype=np.float64)
            for k, p in enumerate(self.plans):
                factors[k, : p.num_edges - 1] = 1.0 if p._scale_factor is None else p._scale_factor
        return slices, factors

    def __call__(self, fine_bej) -> np.ndarray:
        if self.single:
            return fine_bej[:, 0][np.newaxis]  # a view: [1, B, J]
        num_bins, _, num_jobs = fine_bej.shape
        out = np.empty((self.num_scales, num_bins, num_jobs), dtype=np.float64)
        if self.uniform:
            block = fine_bej[:, : self.num_fine]
            if self.scale_factors is not None:
                block = block * self.scale_factors
            for s, (lo, hi) in enumerate(self.slices):
                out[s] = block[:, lo:hi].sum(axis=1)
            return out
        for k, plan in enumerate(self.plans):
            out[:, k] = plan.combine(fine_bej[k, : plan.num_edges - 1].T).T
        return out


def combine_all(plans, fine_bej) -> np.ndarray:
    """One-off form of :class:`CombinePlan` (``PatchLinkage`` keeps the prepared object)."""
    return CombinePlan(plans)(fine_bej)


def threshold_table(plans) -> np.ndarray:
    """f64[B, Emax]. Bins with fewer edges are padded by repeating their last threshold: the
    padded fine bins (t < s <= t) are empty by construction."""
    n_edges = max(p.num_edges for p in plans)
    table = np.empty((len(plans), n_edges), dtype=np.float64)
    for k, plan in enumerate(plans):
        table[k, : plan.num_edges] = plan.thresholds
        table[k, plan.num_edges :] = plan.thresholds[-1]
    return table


FORCE_DEVICE_REDUCE = False  # tests: take the device-resident reduce of the nccl route in any group

JOB_FIXED_COST = 2.0e5  # evaluated-pair equivalent of touching a job at all (launch share, empty windows)


def job_costs(layout1, layout2, jobs, thresholds, tile: int = 1024) -> np.ndarray:
    """Host-side estimate of the device work per job (no GPU involved). ``PatchLinkage.count_pairs``
    balances the exact figure from the device instead (``engine.job_work``); this one serves the CPU
    tests of the sharding logic and as a documented model of the culling.

    Brute force would cost N1*N2; the z-window culling of the device path only evaluates the part
    of patch 1 within (tile extent + 2 r_max) in z of each lane tile of patch 2, so a job costs about
    N2 * N1 * min(1, (zext2 * tile / N2 + 2 r_max) / zext1)."""
    n1 = layout1.segment_sizes()[jobs[:, 0]].sum(axis=1, dtype=np.float64)
    n2 = layout2.segment_sizes()[jobs[:, 1]].sum(axis=1, dtype=np.float64)
    r_max = float(np.sqrt(np.max(thresholds)))
    z1 = np.maximum(layout1.z_extent[jobs[:, 0]], 1e-12)
    z2 = layout2.z_extent[jobs[:, 1]]
    window = np.minimum(1.0, (z2 * np.minimum(1.0, tile / np.maximum(n2, 1.0)) + 2.0 * r_max) / z1)
    return n1 * n2 * window + 1e3 * (n1 + n2)  # + per-object cost of touching the patches at all


def best_sort_axis(center_xyz, num_records) -> int:
    """Coordinate axis (0 = x, 1 = y, 2 = z) most perpendicular to the footprint's mean direction:
    patches are widest along it, so the device's 1-d window culling removes the most pairs. A
    footprint without a preferred direction (full sky) keeps z."""
    mean = (center_xyz * num_records[:, np.newaxis]).sum(axis=0) / max(float(num_records.sum()), 1.0)
    if np.linalg.norm(mean) < 0.2:
        return 2
    return int(np.argmin(np.abs(mean)))


_PLAN_CACHE: list = []  # (config, plans, thresholds, combine) of the last few configurations (they are immutable)


def _plans_for(config):
    """Angular plans, threshold table and recombination plan of a configuration; every measurement with the
    same configuration object (each builds its own PatchLinkage) shares them."""
    for cfg, plans, thresholds, combine in _PLAN_CACHE:
        if cfg is config:
            return plans, thresholds, combine
    plans = angular_plans(config)
    thresholds = threshold_table(plans)
    combine = CombinePlan(plans)
    _PLAN_CACHE.append((config, plans, thresholds, combine))
    del _PLAN_CACHE[:-4]
    return plans, thresholds, combine


class PatchLinkage:
    """Which patch pairs can contain pairs of objects within the largest scale.

    Patches i, j are linked if their centres are closer than r_i + r_j + theta_max
    (measurements.py:193-237); every pair count of one measurement shares the linkage."""

    def __init__(self, config, patch_links: dict) -> None:
        self.config = config
        self.patch_links = patch_links
        self.last_stats = None
        self.sort_axis = 2  # coordinate the device sorts by for its window culling (set by from_catalogs)
        # derived once per linkage (the configuration is immutable): thresholds and job tables
        self._plans = None
        self._thresholds = None
        self._combine = None
        self._job_tables: dict = {}
        self._scatter: dict = {}
        self._partitions: dict = {}
        self.last_rank_info: dict | None = None
        self.last_batch_stats: dict = {}  # count_pairs_batch: CountStats of every count of the last submission, by its name
        self._dense_spec = None

    def _angular_setup(self):
        if self._plans is None:
            self._plans, self._thresholds, self._combine = _plans_for(self.config)
        return self._plans, self._thresholds

    @classmethod
    def from_catalogs(cls, config, catalog: Catalog, *catalogs: Catalog):
        if any(set(cat.keys()) != set(catalog.keys()) for cat in catalogs):
            raise InconsistentPatchesError("patch IDs do not match")
        max_angle = get_max_angle(config).data[0]
        # the catalogue with the most records constrains centres / radii best (measurements.py:220-225)
        ranked = sorted([catalog, *catalogs], key=lambda cat: cat.get_num_records(), reverse=True)
        ref_cat, others = ranked[0], ranked[1:]
        check_patch_conistency(ref_cat, *others)
        patch_ids = list(ref_cat.keys())
        centers, radii = ref_cat.get_centers(), ref_cat.get_radii().data
        # all P x P centre separations at once, with the arithmetic of AngularCoordinates.distance
        # (squares summed over x, y, z, sqrt, 2 asin(r/2)), so the comparison decides exactly as the per-patch loop
        xyz = centers.to_3d()
        chord = np.sqrt(((xyz[:, np.newaxis, :] - xyz[np.newaxis, :, :]) ** 2).sum(axis=2))
        linked = 2.0 * np.arcsin(chord / 2.0) < (radii[np.newaxis, :] + radii[:, np.newaxis] + max_angle)
        links = {pid: set(compress(patch_ids, row)) for pid, row in zip(patch_ids, linked)}
        new = cls(config, links)
        new.sort_axis = best_sort_axis(centers.to_3d(), np.asarray(ref_cat.get_num_records(), dtype=np.float64))
        return new

    @property
    def num_total(self) -> int:
        return len(self.patch_links) ** 2

    @property
    def num_links(self) -> int:
        return sum(len(v) for v in self.patch_links.values())

    @property
    def density(self) -> float:
        return self.num_links / self.num_total

    def __repr__(self) -> str:
        return f"{type(self).__name__}(num_links={self.num_links}, density={self.density:.0%})"

    def iter_patch_id_pairs(self, *, auto: bool):
        """All linked (i, j): every ordered pair for a cross count, i <= j for an auto count
        (measurements.py:258-289). Same set as the reference; the order (diagonal jobs first, then
        by patch id) only matters for scheduling, which happens in ``partition_jobs``."""
        for i in sorted(self.patch_links):
            if i in self.patch_links[i]:
                yield (i, i)
        for i in sorted(self.patch_links):
            for j in sorted(self.patch_links[i]):
                if j != i and (not auto or j > i):
                    yield (i, j)

    def get_patch_pairs(self, catalog1: Catalog, catalog2: Catalog | None = None) -> np.ndarray:
        """int32[n_jobs, 2] job table (stands in for the tuple of ``PatchPair`` objects,
        measurements.py:291-305)."""
        auto = catalog2 is None
        if auto not in self._job_tables:
            pairs = list(self.iter_patch_id_pairs(auto=auto))
            table = np.array(pairs, dtype=np.int32).reshape(-1, 2)
            table.setflags(write=False)
            self._job_tables[auto] = table
        return self._job_tables[auto]

    # ------------------------------------------------------------------ the hot path
    def count_pairs(self, main_catalog: Catalog, *optional_catalog: Catalog, progress: bool = False,
                    max_workers: int | None = None, mode: str = "nn", count_type_info: str | None = None):
        """Pair counts between the patches of one (auto) or two catalogues -> one
        ``NormalisedCounts`` per scale (measurements.py:307-367)."""
        if str(mode) != "nn":
            raise NotImplementedError("only the 'nn' counting mode is part of this build")
        if len(optional_catalog) > 1:
            raise TypeError("count_pairs() takes at most two catalogues")
        if count_type_info is not None:
            _log_info("counting %s from patch pairs", count_type_info)
        auto = len(optional_catalog) == 0
        cat2 = main_catalog if auto else optional_catalog[0]
        binning = self.config.binning.binning
        num_bins, num_patches = len(binning), len(main_catalog)
        layout1, layout2 = _active_layout(main_catalog, num_bins), _active_layout(cat2, num_bins)

        jobs = self.get_patch_pairs(main_catalog, None if auto else cat2)
        plans, thresholds = self._angular_setup()
        num_fine = thresholds.shape[1] - 1

        # Several GPUs, two ways. One process (the drop-in case): the library splits the job list over the devices of
        # its context, ``max_workers`` caps how many (the reference's worker pool, measurements.py:344-350). One process
        # per GPU under torch.distributed: the jobs are sharded over the ranks here and one sum all-reduce combines
        # the result (every slot is non-zero on exactly one rank: the sum is exact and order independent).
        rank, size = parallel.world()
        if size == 1 and not FORCE_DEVICE_REDUCE:
            # one process: the library returns the result tensor [S, B, P, P] from ONE call (yawhip_count_pairs_dense:
            # counting on the GPU(s), then separation weights, per-scale sums, the x 0.5 of an autocorrelation's diagonal
            # jobs and the scatter into the slots in C)
            if self._dense_spec is None:
                self._dense_spec = self._combine.dense_spec(num_fine)
            slices, factors = self._dense_spec
            counts, stats = engine.count_dense(layout1, layout2, jobs, thresholds, slices, factors, auto,
                                               sort_axis=self.sort_axis, max_workers=max_workers)
            self.last_stats = stats
            self._report(count_type_info, len(jobs), stats, progress)
            scale_counts = [PatchedCounts(binning, counts[s], auto=auto) for s in range(counts.shape[0])]
            sum_weights = PatchedSumWeights(binning, layout1.sum_weights_for(num_bins), layout2.sum_weights_for(num_bins),
                                            auto=auto)
            return [NormalisedCounts(c, sum_weights) for c in scale_counts]
        # Several ranks: balance what the device will really evaluate (lane tile x window sizes, from the item builder); the
        # partition is a plan: rank 0 derives it once per (catalogue pair, thresholds, group size) and broadcasts it.
        # The key is rank independent (every rank enters the broadcast below, or none does) and identifies the catalogue pair
        # by content that every rank shares -- per-patch sizes and the sum of weights, not an object id -- and the thresholds.
        import hashlib
        import time as _time

        devices = engine.default_devices(max_workers)  # the devices of the context the counts below run on
        group_device = devices[0]
        timings = dict(partition_ms=0.0, allreduce_ms=0.0, copy_back_ms=0.0)
        digest = hashlib.blake2b(digest_size=16)
        for part in (layout1.offsets, layout2.offsets, thresholds, jobs):
            digest.update(np.ascontiguousarray(part).tobytes())
        key = (digest.hexdigest(), layout1.w is not None, layout2.w is not None, auto, size)
        if key not in self._partitions:
            t_part = _time.perf_counter()
            parts = None
            if rank == 0:
                try:
                    work = engine.job_work(layout1, layout2, jobs, thresholds, sort_axis=self.sort_axis)
                    parts = parallel.partition_jobs(work.astype(np.float64) + JOB_FIXED_COST, size)
                except Exception as err:  # noqa: BLE001 -- the other ranks wait in the broadcast: tell them
                    parts = f"{type(err).__name__}: {err}"  # (as text: an exception object may not pickle)
            parts = parallel.broadcast_object(parts, device=group_device)
            if isinstance(parts, str):
                raise RuntimeError(f"the job partition could not be derived on rank 0: {parts}")
            if len(self._partitions) > 32:
                self._partitions.clear()
            self._partitions[key] = parts
            timings["partition_ms"] = (_time.perf_counter() - t_part) * 1e3
        mine = self._partitions[key][rank]
        failure, fine, rows = None, None, None
        n_compact = len(jobs) * num_bins * num_fine + 1
        # the device-resident route needs ONE device per rank; a context of several devices inside a group (YAW_AMD_DEVICES
        # naming several ids) counts with the library's in-process split and reduces through the host
        on_device = (parallel.device_collectives() or FORCE_DEVICE_REDUCE) and len(devices) == 1
        try:
            if on_device:  # this rank's rows stay in HBM, in their place of the full tensor (zero elsewhere)
                rows, stats = engine.count_rows_device(layout1, layout2, jobs[mine], thresholds, len(jobs), mine,
                                                       sort_axis=self.sort_axis)
            else:
                fine, stats = engine.count_fine(layout1, layout2, jobs[mine], thresholds, sort_axis=self.sort_axis,
                                                max_workers=max_workers)
            self.last_stats = stats
        except Exception as err:  # noqa: BLE001 -- with several ranks the others must not wait for this one forever
            failure = err

        id1, id2 = jobs[:, 0], jobs[:, 1]
        # only linked patch pairs carry counts: the tensor travels in its compact [jobs, B, E-1] form (every rank holds
        # the same job table), 9x smaller than the dense [B, E-1, P, P] at 64 patches; every row is non-zero on exactly
        # one rank, so ONE sum all-reduce (RCCL over xGMI) yields the complete tensor, exactly; one extra element carries
        # the number of ranks that failed, so that all of them raise instead of one leaving the rest blocked in the collective
        if on_device:
            compact = parallel.allreduce_device_rows(rows, n_compact, status=1.0 if failure is not None else 0.0,
                                                     device=group_device, timings=timings)
        else:
            compact = np.zeros(n_compact, dtype=np.float64)
            if failure is not None:
                compact[-1] = 1.0
            elif len(mine):
                compact[:-1].reshape(len(jobs), num_bins, num_fine)[mine] = fine
            t_red = _time.perf_counter()
            compact = parallel.allreduce_sum(compact, device=group_device)
            timings["allreduce_ms"] = (_time.perf_counter() - t_red) * 1e3
        # what a rank's call consisted of, for bench.py's self-verifying multi-GPU line (rank-local; gathered there)
        self.last_rank_info = dict(rank=rank, device=group_device, jobs=int(len(mine)), route="device" if on_device else "host",
                                   count_kernel_ms=float(self.last_stats.count_ms) if failure is None and self.last_stats else None,
                                   **timings)
        if compact[-1] > 0:
            raise RuntimeError(f"pair counting failed on {int(compact[-1])} of {size} ranks") from failure
        fine_bej = np.moveaxis(compact[:-1].reshape(len(jobs), num_bins, num_fine), 0, -1)

        # host epilogue, O(jobs * B * E): separation weights, per-scale recombination, halving of the doubly
        # counted diagonal of an autocorrelation (trees.py:358-362, measurements.py:361-364), scatter into [B, P, P]
        num_scales = self.config.scales.num_scales
        per_scale = self._combine(fine_bej)  # [S, B, n_jobs]
        skey = (auto, num_patches)
        if skey not in self._scatter:  # position of every job in a [P, P] slice and the diagonal factor
            flat = np.ascontiguousarray(id1.astype(np.int64) * num_patches + id2)
            halve = np.where(id1 == id2, 0.5, 1.0) if auto else None
            self._scatter[skey] = (flat, halve)
        flat, halve = self._scatter[skey]
        # [S, B, jobs] -> [S, B, i * P + j], zero elsewhere (one pass in the library: yawhip_host_scatter_rows)
        counts = _lib.scatter_rows((num_scales, num_bins, num_patches, num_patches), flat, per_scale, halve)
        scale_counts = [PatchedCounts(binning, counts[s], auto=auto) for s in range(num_scales)]

        sum_weights = PatchedSumWeights(binning, layout1.sum_weights_for(num_bins), layout2.sum_weights_for(num_bins),
                                        auto=auto)
        if failure is None and self.last_stats is not None:
            self._report(count_type_info, len(jobs), self.last_stats, progress)
        return [NormalisedCounts(counts, sum_weights) for counts in scale_counts]

    def count_pairs_batch(self, requests, *, progress: bool = False, max_workers: int | None = None) -> list:
        """The pair counts of ONE measurement -- ``requests`` = ``[(catalogs, info), ...]`` with ``catalogs`` a tuple of one
        (auto count) or two catalogues, e.g. DD, DR, RD, RR of ``crosscorrelate``
        (src/yaw/correlation/measurements.py:617-628) -- as one submission: every count is put on the GPU's stream at once
        (``yawhip_count_pairs_dense_batch``), the host prepares count k + 1 and writes the tensor of count k while the
        device counts. Returns what ``count_pairs`` returns for each request, in order (``None`` per scale for a request
        with a missing catalogue, as ``count_pairs_optional``). With several ranks, or when the device-resident reduce is
        forced, the counts run one after the other through ``count_pairs``."""
        num_scales = self.config.scales.num_scales
        rank, size = parallel.world()
        results: list = [[None] * num_scales for _ in requests]
        todo = []
        for i, (catalogs, info) in enumerate(requests):
            if len(catalogs) not in (1, 2):
                raise TypeError("a count takes one or two catalogues")
            if any(cat is None for cat in catalogs):
                continue  # (count_pairs_optional: a missing random sample)
            todo.append((i, catalogs[0], catalogs[1] if len(catalogs) == 2 else None, info))
        if size > 1 or FORCE_DEVICE_REDUCE or len(todo) <= 1:
            for i, main, other, info in todo:
                args = (main,) if other is None else (main, other)
                results[i] = self.count_pairs(*args, progress=progress, max_workers=max_workers, count_type_info=info)
            return results
        binning = self.config.binning.binning
        num_bins = len(binning)
        plans, thresholds = self._angular_setup()
        if self._dense_spec is None:
            self._dense_spec = self._combine.dense_spec(thresholds.shape[1] - 1)
        slices, factors = self._dense_spec
        pairs, meta = [], []
        for i, main, other, info in todo:
            auto = other is None
            cat2 = main if auto else other
            layout1, layout2 = _active_layout(main, num_bins), _active_layout(cat2, num_bins)
            jobs = self.get_patch_pairs(main, None if auto else cat2)
            if info is not None:
                _log_info("counting %s from patch pairs", info)
            pairs.append((layout1, layout2, jobs, auto))
            meta.append((i, layout1, layout2, auto, info, len(jobs)))
        outs = engine.count_dense_batch(pairs, thresholds, slices, factors, sort_axis=self.sort_axis, max_workers=max_workers)
        self.last_batch_stats = {}
        for (i, layout1, layout2, auto, info, n_jobs), (counts, stats) in zip(meta, outs):
            self.last_stats = stats
            self.last_batch_stats[info or str(i)] = stats
            self._report(info, n_jobs, stats, progress)
            scale_counts = [PatchedCounts(binning, counts[s], auto=auto) for s in range(counts.shape[0])]
            sum_weights = PatchedSumWeights(binning, layout1.sum_weights_for(num_bins), layout2.sum_weights_for(num_bins), auto=auto)
            results[i] = [NormalisedCounts(c, sum_weights) for c in scale_counts]
        return results

    @staticmethod
    def _report(what, n_jobs, stats, progress) -> None:
        """The reference logs every pair count and shows a progress bar over its patch-pair tasks
        (src/yaw/correlation/measurements.py:53-62,344-350); one GPU call has no tasks to tick off, so ``progress`` prints
        one summary line per count instead. Rank 0 only."""
        if not progress and not logger.isEnabledFor(logging.INFO):
            return  # nobody listens: the line is not even formatted (a pair count can be a 0.5 ms call)
        if parallel.world()[0] != 0:
            return
        secs = max(stats.total_ms, 1e-6) / 1e3
        line = (f"{what or 'pair count'}: {n_jobs} patch pairs, {stats.candidate_pairs:.4g} candidate pairs in "
                f"{stats.total_ms:.2f} ms ({stats.candidate_pairs / secs:.3g} pairs/s)")
        logger.info(line)
        if progress:
            print(line, flush=True)

    def count_pairs_optional(self, main_catalog, *optional_catalog, **kwargs):
        """``count_pairs`` that yields ``None`` per scale if any catalogue is missing
        (measurements.py:369-392)."""
        if any(cat is None for cat in (main_catalog, *optional_catalog)):
            return [None] * self.config.scales.num_scales
        return self.count_pairs(main_catalog, *optional_catalog, **kwargs)


def _active_layout(catalog: Catalog, num_bins: int):
    layout = catalog._active_layout
    if layout is None:  # the reference fails in BinnedTrees.__init__ (trees.py:473-474)
        raise FileNotFoundError("no trees found for catalog: call build_trees() first")
    if layout.num_bins not in (1, num_bins):
        raise ValueError(f"catalog was binned into {layout.num_bins} redshift bins, configuration has {num_bins}")
    return layout


def _require_distinct(*catalogs) -> None:
    """Counterpart of ``ensure_unique_catalogs`` (measurements.py:432-448): the reference rejects
    catalogues that share a cache directory; here the state that must not be shared is the active
    layout of a catalogue object."""
    present = [cat for cat in catalogs if cat is not None]
    if len({id(cat) for cat in present}) != len(present):
        raise ValueError("each catalog must be a separate Catalog instance to avoid interference.")


def autocorrelate(config, data: Catalog, random: Catalog, *, count_rr: bool = True, progress: bool = False,
                  max_workers: int | None = None) -> list:
    """Angular autocorrelation in redshift slices: DD, DR and (optionally) RR -> ``[CorrFunc]``, one
    per scale (measurements.py:455-525)."""
    _require_distinct(data, random)
    edges, closed = config.binning.edges, config.binning.closed
    _log_info("building data trees")
    data.build_trees(edges, closed=closed)
    _log_info("building random trees")
    random.build_trees(edges, closed=closed)
    _log_info("computing auto-correlation from DD, DR" + (", RR" if count_rr else ""))
    links = PatchLinkage.from_catalogs(config, data, random)
    # the reference issues DD, DR, RR one after the other (measurements.py:517-523); here they are ONE submission
    DD, DR, RR = links.count_pairs_batch(
        [((data,), "DD"), ((data, random), "DR"), ((random if count_rr else None,), "RR")],
        progress=progress, max_workers=max_workers)
    return [CorrFunc(dd, dr, None, rr) for dd, dr, rr in zip(DD, DR, RR)]


def crosscorrelate(config, reference: Catalog, unknown: Catalog, *, ref_rand: Catalog | None = None,
                   unk_rand: Catalog | None = None, progress: bool = False, max_workers: int | None = None) -> list:
    """Angular cross-correlation between redshift slices of ``reference`` and the whole ``unknown``
    sample: DD always, DR / RD / RR depending on the randoms given -> ``[CorrFunc]``, one per scale
    (measurements.py:529-628)."""
    _require_distinct(reference, unknown, ref_rand, unk_rand)
    count_dr, count_rd = unk_rand is not None, ref_rand is not None
    if not count_dr and not count_rd:
        raise ValueError("at least one random dataset must be provided")
    edges, closed = config.binning.edges, config.binning.closed
    randoms = []
    _log_info("building reference data trees")
    reference.build_trees(edges, closed=closed)
    if count_rd:
        ref_rand.build_trees(edges, closed=closed)
        randoms.append(ref_rand)
    unknown.build_trees(None)
    if count_dr:
        unk_rand.build_trees(None)
        randoms.append(unk_rand)
    _log_info("computing cross-correlation from DD" + (", DR" if count_dr else "") + (", RD" if count_rd else "")
              + (", RR" if count_dr and count_rd else ""))
    links = PatchLinkage.from_catalogs(config, reference, unknown, *randoms)
    # the reference issues DD, DR, RD, RR one after the other (measurements.py:617-628); here they are ONE submission
    DD, DR, RD, RR = links.count_pairs_batch(
        [((reference, unknown), "DD"), ((reference, unk_rand), "DR"), ((ref_rand, unknown), "RD"), ((ref_rand, unk_rand), "RR")],
        progress=progress, max_workers=max_workers)
    return [CorrFunc(dd, dr, rd, rr) for dd, dr, rd, rr in zip(DD, DR, RD, RR)]
