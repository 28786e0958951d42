"""Dense per-(bin, patch, patch) result containers and their jackknife patch sums.

Mirror of ``yaw.correlation.paircounts`` (src/yaw/correlation/paircounts.py:46-616): same
attribute names and array layouts (``counts[B,P,P]``, ``sum_weights1/2[B,P]``), same leave-one-out
arithmetic; HDF5 serialisation is out of scope.
"""
from __future__ import annotations

import numpy as np

from .corrdata import SampledData

__all__ = ["PatchedSumWeights", "PatchedCounts", "NormalisedCounts"]


def _as_index(item):
    return [item] if isinstance(item, (int, np.integer)) else item


class _BinPatchArray:
    """Shared behaviour: anything that can be viewed as f64[B, P, P]."""

    __slots__ = ()

    @property
    def num_bins(self) -> int:
        return len(self.binning)

    def __repr__(self) -> str:
        return f"{type(self).__name__}(auto={self.auto}, binning={self.binning}, num_patches={self.num_patches})"

    def is_compatible(self, other, *, require: bool = False) -> bool:
        if not isinstance(other, type(self)):
            if require:
                raise TypeError(f"{type(other)} is not compatible with {type(self)}")
            return False
        if self.binning != other.binning:
            if require:
                raise ValueError("binning does not match")
            return False
        if self.num_patches != other.num_patches:
            if require:
                raise ValueError("number of patches does not match")
            return False
        return True

    def sample_patch_sum(self) -> SampledData:
        """Sum over all patch pairs and the P leave-one-out sums (paircounts.py:113-141):
        sample_i = total - (row i) - (column i) + (diagonal element i)."""
        arr = self.get_array()
        total = arr.sum(axis=(1, 2))
        along_rows = arr.sum(axis=2).T      # [P,B]: pairs whose first patch is i
        along_cols = arr.sum(axis=1).T      # [P,B]: pairs whose second patch is i
        diagonal = np.diagonal(arr, axis1=1, axis2=2).T
        samples = total[np.newaxis, :] - along_cols - along_rows + diagonal
        return SampledData(self.binning, total, samples)


class PatchedSumWeights(_BinPatchArray):
    """Sum of weights per (bin, patch) of both catalogues (paircounts.py:144-288)."""

    __slots__ = ("binning", "auto", "sum_weights1", "sum_weights2")

    def __init__(self, binning, sum_weights1, sum_weights2, *, auto: bool) -> None:
        self.binning = binning
        self.auto = bool(auto)
        sum_weights1, sum_weights2 = np.asarray(sum_weights1), np.asarray(sum_weights2)
        if sum_weights1.ndim != 2 or sum_weights2.ndim != 2:
            raise ValueError("'sum_weights1/2' must be two-dimensional")
        if sum_weights1.shape != sum_weights2.shape:
            raise ValueError("'sum_weights1' and 'sum_weights2' must have the same shape")
        if sum_weights1.shape[0] != len(binning):
            raise ValueError("first dimension of 'sum_weights1/2' must match 'binning'")
        self.sum_weights1 = sum_weights1.astype(np.float64)
        self.sum_weights2 = sum_weights2.astype(np.float64)

    @property
    def num_patches(self) -> int:
        return self.sum_weights1.shape[1]

    def __eq__(self, other) -> bool:
        if not isinstance(other, type(self)):
            return NotImplemented
        return (
            self.binning == other.binning
            and self.auto == other.auto
            and np.array_equal(self.sum_weights1, other.sum_weights1)
            and np.array_equal(self.sum_weights2, other.sum_weights2)
        )

    def get_array(self):
        """Outer product per bin; an autocorrelation keeps the upper triangle and half the
        diagonal (paircounts.py:267-288)."""
        arr = self.sum_weights1[:, :, np.newaxis] * self.sum_weights2[:, np.newaxis, :]
        if self.auto:
            arr = np.triu(arr)
            idx = np.arange(self.num_patches)
            arr[:, idx, idx] *= 0.5
        return arr

    def bins_subset(self, item):
        item = _as_index(item)
        return type(self)(self.binning[item], self.sum_weights1[item], self.sum_weights2[item], auto=self.auto)

    def patches_subset(self, item):
        item = _as_index(item)
        return type(self)(self.binning, self.sum_weights1[:, item], self.sum_weights2[:, item], auto=self.auto)


class PatchedCounts(_BinPatchArray):
    """Pair counts per (bin, patch 1, patch 2) (paircounts.py:291-456)."""

    __slots__ = ("binning", "counts", "auto")

    def __init__(self, binning, counts, *, auto: bool) -> None:
        self.binning = binning
        self.auto = bool(auto)
        counts = np.asarray(counts)
        if counts.ndim != 3:
            raise ValueError("'counts' must be three-dimensional")
        if counts.shape[0] != len(binning):
            raise ValueError("first dimension of 'counts' must match 'binning'")
        if counts.shape[1] != counts.shape[2]:
            raise ValueError("'counts' must have shape (num_bins, num_patches, num_patches)")
        self.counts = counts.astype(np.float64, copy=False)  # a float64 input is adopted, not copied (4 MB per scale at 128 patches)

    @classmethod
    def zeros(cls, binning, num_patches: int, *, auto: bool):
        return cls(binning, np.zeros((len(binning), num_patches, num_patches)), auto=auto)

    @property
    def num_patches(self) -> int:
        return self.counts.shape[1]

    def __eq__(self, other) -> bool:
        if not isinstance(other, type(self)):
            return NotImplemented
        return self.binning == other.binning and self.auto == other.auto and np.array_equal(self.counts, other.counts)

    def __add__(self, other):
        self.is_compatible(other, require=True)
        return type(self)(self.binning, self.counts + other.counts, auto=self.auto)

    def __mul__(self, factor):
        return type(self)(self.binning, self.counts * factor, auto=self.auto)

    def get_array(self):
        return self.counts

    def set_patch_pair(self, patch_id1: int, patch_id2: int, counts_binned) -> None:
        self.counts[:, patch_id1, patch_id2] = counts_binned

    def bins_subset(self, item):
        item = _as_index(item)
        return type(self)(self.binning[item], self.counts[item], auto=self.auto)

    def patches_subset(self, item):
        item = np.atleast_1d(_as_index(item))
        return type(self)(self.binning, self.counts[:, item][:, :, item], auto=self.auto)


class NormalisedCounts(_BinPatchArray):
    """Pair counts together with their normalisation (paircounts.py:459-616)."""

    __slots__ = ("_counts", "_weights")

    def __init__(self, counts: PatchedCounts, sum_weights: PatchedSumWeights) -> None:
        if counts.num_patches != sum_weights.num_patches:
            raise ValueError("number of patches of counts- and weights-container does not match")
        if counts.num_bins != sum_weights.num_bins:
            raise ValueError("number of bins of counts- and weights-container does not match")
        self._counts = counts
        self._weights = sum_weights

    @property
    def counts(self) -> PatchedCounts:
        return self._counts

    @property
    def sum_weights(self) -> PatchedSumWeights:
        return self._weights

    @property
    def binning(self):
        return self._counts.binning

    @property
    def auto(self) -> bool:
        return self._counts.auto

    @property
    def num_patches(self) -> int:
        return self._counts.num_patches

    def is_compatible(self, other, *, require: bool = False) -> bool:
        if type(self) is not type(other):
            if require:
                raise TypeError(f"{type(other)} is not compatible with {type(self)}")
            return False
        return self._counts.is_compatible(other._counts, require=require)

    def __eq__(self, other) -> bool:
        if type(self) is not type(other):
            return NotImplemented
        return self._counts == other._counts and self._weights == other._weights

    def get_array(self):
        norm = self._weights.sample_patch_sum().data
        return self._counts.get_array() / norm[:, np.newaxis, np.newaxis]

    def sample_patch_sum(self) -> SampledData:
        """counts / (product of weight sums), for the full sample and every jackknife sample
        (paircounts.py:559-565)."""
        c, w = self._counts.sample_patch_sum(), self._weights.sample_patch_sum()
        return SampledData(self.binning, c.data / w.data, c.samples / w.samples)

    def bins_subset(self, item):
        return type(self)(self._counts.bins_subset(item), self._weights.bins_subset(item))

    def patches_subset(self, item):
        return type(self)(self._counts.patches_subset(item), self._weights.patches_subset(item))
