"""Host side of the per-tree count: angular limits -> chord thresholds, and the recombination of
the fine-bin counts the device returns into one value per correlation scale.

Everything here is O(B * E) per measurement.  It follows the reference's
``AngularTree.count`` (src/yaw/catalog/trees.py:303-362) step by step because the *thresholds
actually used* are part of the bit-parity contract (SURVEY.md findings 1 and 2):

  limits  --log10--> unique --10**--> ang_bins --2 sin(x/2)--> r --libm pow(r, 2.0)--> t

The device only ever sees ``t``.
"""
from __future__ import annotations

import math

import numpy as np

__all__ = ["AngularBinPlan", "plan_for_limits", "parse_ang_limits", "get_ang_bins", "logarithmic_mid",
           "get_counts_for_limits", "chord_thresholds"]


def parse_ang_limits(ang_min, ang_max):
    """Validate and stack lower / upper limits into f64[S, 2] (trees.py:46-81)."""
    lo = np.atleast_1d(ang_min).astype(np.float64)
    hi = np.atleast_1d(ang_max).astype(np.float64)
    if lo.ndim != 1 or hi.ndim != 1:
        raise ValueError("'ang_min' and 'ang_max' must be 1-dim")
    if len(lo) != len(hi):
        raise ValueError("length of 'ang_min' and 'ang_max' does not match")
    if np.any(lo >= hi):
        raise ValueError("'ang_min' < 'ang_max' not satisfied")
    limits = np.column_stack((lo, hi))
    if np.any(limits < 0.0) or np.any(limits > np.pi):
        raise ValueError("'ang_min' and 'ang_max' not in range [0.0, pi]")
    return limits


def get_ang_bins(ang_range, weight_scale, weight_res):
    """Unique, sorted angular bin edges; with ``weight_scale`` a log-spaced refinement is merged
    in (trees.py:84-117).  The round trip through log10 / 10** is deliberate."""
    with np.errstate(divide="ignore"):
        log_range = np.log10(ang_range)
    pieces = [log_range.flatten()]
    if weight_scale is not None:
        pieces.insert(0, np.linspace(log_range.min(), log_range.max(), weight_res + 1))
    return 10.0 ** np.sort(np.unique(np.concatenate(pieces)))


def logarithmic_mid(edges):
    """Logarithmic bin centres (trees.py:120-124)."""
    log_edges = np.log10(edges)
    return 10.0 ** ((log_edges[:-1] + log_edges[1:]) / 2.0)


def get_counts_for_limits(counts, ang_bins, ang_limits):
    """Sum the fine bins between the edges nearest to each scale's limits (trees.py:134-160)."""
    out = np.empty(len(ang_limits), dtype=counts.dtype)
    for s, (lo, hi) in enumerate(ang_limits):
        first = np.argmin(np.abs(ang_bins - lo))
        last = np.argmin(np.abs(ang_bins - hi))
        out[s] = counts[first:last].sum()
    return out


def chord_thresholds(ang_bins):
    """t = pow(2 sin(theta/2), 2.0) with libm's pow, NOT r*r (they differ for ~0.08 % of radii and
    scipy's count_neighbors compares against pow(r, 2.0): SURVEY.md 8(a11))."""
    r = 2.0 * np.sin(np.asarray(ang_bins, dtype=np.float64) / 2.0)  # coordinates.py:277
    return np.array([math.pow(float(v), 2.0) for v in r], dtype=np.float64)


class AngularBinPlan:
    """Everything derived from one redshift bin's angular limits."""

    __slots__ = ("limits", "ang_bins", "thresholds", "rweight", "_scale_factor", "_slices")

    def __init__(self, limits, ang_bins, rweight):
        self.limits = limits
        self.ang_bins = ang_bins
        self.thresholds = chord_thresholds(ang_bins)
        self.rweight = rweight
        self._scale_factor = None
        if rweight is not None:  # trees.py:358-360
            ang_weights = logarithmic_mid(ang_bins) ** rweight
            self._scale_factor = ang_weights / ang_weights.sum()
        self._slices = [
            (int(np.argmin(np.abs(ang_bins - lo))), int(np.argmin(np.abs(ang_bins - hi)))) for lo, hi in limits
        ]

    @property
    def num_edges(self) -> int:
        return len(self.ang_bins)

    @property
    def num_scales(self) -> int:
        return len(self.limits)

    def combine(self, fine):
        """fine f64[..., E-1] -> f64[..., S]: optional separation weighting, then the per-scale sums."""
        fine = np.asarray(fine, dtype=np.float64)
        if self._scale_factor is not None:
            fine = fine * self._scale_factor
        out = np.empty(fine.shape[:-1] + (self.num_scales,), dtype=np.float64)
        for s, (first, last) in enumerate(self._slices):
            out[..., s] = fine[..., first:last].sum(axis=-1)
        return out


def plan_for_limits(ang_min, ang_max, rweight=None, resolution=None) -> AngularBinPlan:
    limits = parse_ang_limits(ang_min, ang_max)
    if rweight is not None and resolution is None:
        resolution = 50  # AngularTree.count default weight_res (trees.py:310)
    return AngularBinPlan(limits, get_ang_bins(limits, rweight, resolution), rweight)
