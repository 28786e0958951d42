"""Device session: one ``yawhip`` context per process, catalogue layouts uploaded once and kept
resident in HBM, and the call that counts the fine-bin pairs for a list of jobs.

``count_fine`` is the single seam between the host driver (measurements.py) and the HIP library.
It has no CPU fallback; the multi-process CPU tests replace it with the oracle to exercise the
sharding / reduction logic without a GPU.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .parallel import local_device_index

__all__ = ["get_context", "device_catalog", "count_fine", "release", "default_kernel"]

_contexts: dict = {}
default_kernel = "auto"


def get_context(device: int | None = None) -> "_lib.Context":
    device = local_device_index() if device is None else int(device)
    ctx = _contexts.get(device)
    if ctx is None:
        ctx = _contexts[device] = _lib.Context(device)
    return ctx


def release() -> None:
    """Destroy all contexts (device catalogues must have been freed before)."""
    for ctx in _contexts.values():
        ctx.close()
    _contexts.clear()


def device_catalog(layout, ctx=None, sort_axis: int = 2) -> "_lib.DeviceCatalog":
    """Upload (once per context) and return the device copy of a layout. ``sort_axis`` is the
    coordinate the library sorts segments by for its window culling; a copy made for another axis
    is replaced."""
    ctx = ctx or get_context()
    dev = layout.device.get(id(ctx))
    if dev is not None and dev.sort_axis != sort_axis:
        dev.free()
        dev = None
    if dev is None:
        dev = _lib.DeviceCatalog(ctx, layout.x, layout.y, layout.z, layout.w, layout.num_patches, layout.num_bins,
                                 layout.offsets, sort_axis=sort_axis)
        layout.device[id(ctx)] = dev
    return dev


def count_fine(layout1, layout2, jobs, thresholds, *, kernel: str | None = None, sort_axis: int = 2):
    """Fine-bin pair counts for ``jobs`` (int[n,2]) -> (f64[n_jobs, B, E-1], CountStats).

    Unweighted catalogues are counted in int64 on the device and converted exactly
    (the reference's ``.astype(np.float64)``, trees.py:353)."""
    ctx = get_context()
    d1 = device_catalog(layout1, ctx, sort_axis)
    d2 = d1 if layout2 is layout1 else device_catalog(layout2, ctx, sort_axis)
    counts, sums, stats = _lib.count_pairs(ctx, d1, d2, jobs, thresholds, kernel=kernel or default_kernel)
    fine = sums if sums is not None else counts.astype(np.float64)
    return fine, stats
