"""Device session: one ``yawhip`` context per process, catalogue layouts uploaded once and kept
resident in HBM, and the call that counts the fine-bin pairs for a list of jobs.

``count_fine`` is the single seam between the host driver (measurements.py) and the HIP library.
It has no CPU fallback; the multi-process CPU tests replace it with the oracle to exercise the
sharding / reduction logic without a GPU.
"""
from __future__ import annotations

import numpy as np

import os

from . import _lib
from .parallel import local_device_index, world

__all__ = ["get_context", "device_catalog", "count_fine", "count_dense", "count_dense_batch", "job_work", "assign_patches", "release", "default_kernel"]

_contexts: dict = {}
default_kernel = "auto"
forced_strip_micro: int | None = None  # set to pin the strip grid spacing (bench / experiments)


def default_devices(max_workers: int | None = None) -> tuple:
    """GPUs one process counts on. Inside a ``torch.distributed`` group (one process per GPU) that is the process'
    own device; a single process takes every visible GPU (``YAW_AMD_DEVICES="0,1,2"`` picks them explicitly, an id
    may repeat) -- the counterpart of the reference's worker pool, so ``max_workers`` caps their number
    (src/yaw/utils/parallel.py:145-150). Launchers that start one process per GPU without setting LOCAL_RANK (mpirun,
    srun) must set ``YAW_AMD_DEVICE`` (or a one-id ``YAW_AMD_DEVICES``) per process, or every process takes every GPU.
    Inside a group the collectives run on the counting context's device (``Context.device``); a context of SEVERAL
    devices inside a group counts through the host route (``PatchLinkage.count_pairs``)."""
    env = os.environ.get("YAW_AMD_DEVICES")
    if env:
        devices = [int(v) for v in env.split(",") if v.strip() != ""]
    elif world()[1] > 1 or "LOCAL_RANK" in os.environ or "YAW_AMD_DEVICE" in os.environ:
        devices = [local_device_index()]
    else:
        devices = list(range(max(_lib.device_count(), 1)))
    if max_workers is not None and max_workers >= 1:
        devices = devices[: int(max_workers)]
    return tuple(devices)


def get_context(device=None, max_workers: int | None = None) -> "_lib.Context":
    """The context of ``device`` (an id or a sequence of ids), by default of :func:`default_devices`."""
    if device is None:
        device = default_devices(max_workers)
    key = tuple(int(d) for d in device) if isinstance(device, (list, tuple)) else (int(device),)
    ctx = _contexts.get(key)
    if ctx is None:
        ctx = _contexts[key] = _lib.Context(key[0] if len(key) == 1 else list(key))
    return ctx


def release() -> None:
    """Destroy all contexts (device catalogues must have been freed before)."""
    for ctx in _contexts.values():
        ctx.close()
    _contexts.clear()


_strip_micro_cache: dict = {}


def strip_micro_for(thresholds) -> int:
    """Spacing of the strip grid (1e-6 chord units) that suits the widest separation of a threshold
    table: just above the largest chord, so that a run has partners in three strips only and those are
    as narrow as possible. Measured on the 10M x 10M headline (chord 2909): 2950 -> 2.09 ms per step,
    3600 -> 2.16, 4400 -> 2.21, 5200 -> 2.28; below the chord five strips take part (2000: 30 % slower)."""
    key = (id(thresholds), thresholds.shape)
    hit = _strip_micro_cache.get(key)
    if hit is not None and hit[0] is thresholds:  # threshold tables are built once per configuration and never modified
        return hit[1]
    r = float(np.sqrt(np.max(thresholds)))
    micro = int(min(max(np.ceil(1.02e6 * r / 50.0) * 50.0, 1000), 100000))
    if len(_strip_micro_cache) > 16:
        _strip_micro_cache.clear()
    _strip_micro_cache[key] = (thresholds, micro)
    return micro


def device_catalog(layout, ctx=None, sort_axis: int = 2, strip_micro: int | None = None,
                   exact: bool = False) -> "_lib.DeviceCatalog":
    """Upload (once per context) and return the device copy of a layout. ``sort_axis`` is the
    coordinate the library sorts segments by for its window culling; ``strip_micro`` the wanted
    spacing of the strip grid. A copy made for another axis, or for a grid more than 1.6 x off the
    wanted spacing (``exact``: any other spacing), is replaced."""
    ctx = ctx or get_context()
    dev = layout.device.get(id(ctx))
    if dev is not None:
        stale = dev.sort_axis != sort_axis
        if strip_micro is not None and not stale:
            have = dev.strip_micro
            stale = have != strip_micro if (exact or have == 0 or strip_micro == 0) else \
                not (strip_micro / 1.6 <= have <= strip_micro * 1.6)
        if stale:
            dev.free()
            dev = None
    if dev is None:
        dev = _lib.DeviceCatalog(ctx, layout.x, layout.y, layout.z, layout.w, layout.num_patches, layout.num_bins,
                                 layout.offsets, sort_axis=sort_axis, strip_micro=strip_micro)
        layout.device[id(ctx)] = dev
    return dev


def count_fine(layout1, layout2, jobs, thresholds, *, kernel: str | None = None, sort_axis: int = 2,
               max_workers: int | None = None):
    """Fine-bin pair counts for ``jobs`` (int[n,2]) -> (f64[n_jobs, B, E-1], CountStats).

    Unweighted catalogues are counted in int64 on the device and converted exactly
    (the reference's ``.astype(np.float64)``, trees.py:353). With several GPUs in the process' context the library
    splits the jobs over them; ``max_workers`` caps how many are used."""
    ctx, d1, d2 = _device_pair(layout1, layout2, thresholds, sort_axis, max_workers)
    counts, sums, stats = _lib.count_pairs(ctx, d1, d2, jobs, thresholds, kernel=kernel or default_kernel)
    fine = sums if sums is not None else counts.astype(np.float64)
    return fine, stats


def count_dense(layout1, layout2, jobs, thresholds, slices, fine_factors, halve_diagonal, *, kernel: str | None = None,
                sort_axis: int = 2, max_workers: int | None = None):
    """The result tensor f64[S, B, P, P] of one pair count from ONE library call (``yawhip_count_pairs_dense``) and its
    ``CountStats``; ``slices`` / ``fine_factors`` describe the per-scale recombination (``CombinePlan.dense_spec``)."""
    ctx, d1, d2 = _device_pair(layout1, layout2, thresholds, sort_axis, max_workers)
    return _lib.count_pairs_dense(ctx, d1, d2, np.ascontiguousarray(jobs, dtype=np.int32).reshape(-1, 2),
                                  np.ascontiguousarray(thresholds, dtype=np.float64), slices, fine_factors, halve_diagonal,
                                  kernel=kernel or default_kernel)


def count_dense_batch(pairs, thresholds, slices, fine_factors, *, kernel: str | None = None, sort_axis: int = 2,
                      max_workers: int | None = None):
    """Several counts of one measurement from ONE library call (``yawhip_count_pairs_dense_batch``): ``pairs`` is a
    sequence of ``(layout1, layout2, jobs, halve_diagonal)``; all catalogues are uploaded (once) first, then every count is
    put on the stream. Returns ``[(f64[S, B, P, P], CountStats), ...]`` in the order of ``pairs``."""
    requests, ctx = [], None
    for _ in range(3):  # an upload may replace a copy made for another strip grid -- one an earlier pair refers to: look again
        requests = []
        for layout1, layout2, jobs, halve in pairs:
            ctx, d1, d2 = _device_pair(layout1, layout2, thresholds, sort_axis, max_workers)
            requests.append((d1, d2, jobs, halve))
        if all(d1._h and d2._h for d1, d2, _, _ in requests):
            break
    else:
        raise _lib.YawhipError("count_dense_batch: the catalogues of the batch do not settle on one strip grid")
    if not requests:
        return []
    return _lib.count_pairs_dense_batch(ctx, requests, np.ascontiguousarray(thresholds, dtype=np.float64), slices, fine_factors,
                                        kernel=kernel or default_kernel)


def count_rows_device(layout1, layout2, jobs, thresholds, n_rows_total: int, row_index, *, kernel: str | None = None,
                      sort_axis: int = 2):
    """This rank's share of a sharded count, left on the device in its place of the full [jobs, B, E-1] tensor
    (``yawhip_count_pairs_rows_device``) -> (``_lib.DeviceRows``, CountStats)."""
    ctx, d1, d2 = _device_pair(layout1, layout2, thresholds, sort_axis)
    return _lib.count_pairs_rows_device(ctx, d1, d2, jobs, thresholds, n_rows_total, row_index, kernel=kernel or default_kernel)


def _device_pair(layout1, layout2, thresholds, sort_axis, max_workers=None):
    ctx = get_context(max_workers=max_workers)
    micro = forced_strip_micro if forced_strip_micro is not None else strip_micro_for(thresholds)
    d1 = device_catalog(layout1, ctx, sort_axis, micro, exact=forced_strip_micro is not None)
    d2 = d1 if layout2 is layout1 else device_catalog(layout2, ctx, sort_axis, d1.strip_micro, exact=True)
    return ctx, d1, d2


def job_work(layout1, layout2, jobs, thresholds, *, kernel: str | None = None, sort_axis: int = 2) -> np.ndarray:
    """Pair distances the device will evaluate for every job (int64[n_jobs]; ``yawhip_job_work``): the
    cost ``PatchLinkage.count_pairs`` balances when it shards the job list over GPUs. It is an exact
    function of the inputs, so every rank derives the same partition."""
    ctx, d1, d2 = _device_pair(layout1, layout2, thresholds, sort_axis)
    return _lib.job_work(ctx, d1, d2, jobs, thresholds, kernel=kernel or default_kernel)


def assign_patches(xyz, centers_xyz):
    """Nearest patch centre per object on the device (``yawhip_assign_patches``), or ``None`` when no
    GPU / library is available -- patch assignment is catalogue preparation, which the reference does on
    the host too, so unlike the pair counts it may fall back to scipy there."""
    try:
        if _lib.device_count() < 1:
            return None
        ctx = get_context(default_devices()[0])  # one device does it: no reason to span (and replicate on) every GPU
    except _lib.YawhipError:
        return None
    if isinstance(xyz, tuple):  # three columns
        x, y, z = (np.ascontiguousarray(c, dtype=np.float64) for c in xyz)
    else:
        xyz = np.asarray(xyz, dtype=np.float64)
        x, y, z = (np.ascontiguousarray(xyz[:, a]) for a in range(3))
    ids = _lib.assign_patches(ctx, x, y, z, centers_xyz)
    return ids.astype(np.int64)
