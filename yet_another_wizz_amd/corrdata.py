"""Binned values with jackknife samples (mirror of ``yaw.correlation.corrdata``,
src/yaw/correlation/corrdata.py:48-260,383-; ASCII / plotting I/O is out of scope)."""
from __future__ import annotations

import warnings

import numpy as np

__all__ = ["SampledData", "CorrData", "cov_from_samples"]


def cov_from_samples(samples, rowvar: bool = False, kind: str = "full"):
    """Jackknife covariance: np.cov(ddof=0) * (M - 1) (corrdata.py:48-106)."""
    if kind not in ("full", "diag", "var"):
        raise ValueError(f"invalid covariance kind '{kind}'")
    ax_samples, ax_observ = (1, 0) if rowvar else (0, 1)
    blocks = None
    if isinstance(samples, np.ndarray) and samples.ndim == 2:
        joint = samples
    else:
        blocks = [np.asarray(s) for s in samples]
        joint = np.concatenate(blocks, axis=ax_observ)
    n_samples, n_observ = joint.shape[ax_samples], joint.shape[ax_observ]
    if n_samples == 1:
        return np.full((n_observ, n_observ), np.nan)
    cov = np.cov(joint, rowvar=rowvar, ddof=0) * (n_samples - 1)
    cov = np.atleast_2d(cov)
    if kind == "var":
        cov = np.diag(np.diag(cov))
    elif kind == "diag":
        keep = np.diag(np.diag(cov))
        shift = 0
        for block in blocks or []:
            shift += block.shape[ax_observ]
            if shift >= n_observ:
                break
            keep += np.diag(np.diag(cov, k=-shift), k=-shift) + np.diag(np.diag(cov, k=shift), k=shift)
        cov = keep
    return cov


class SampledData:
    """Values in B redshift bins plus M jackknife realisations (corrdata.py:109-260)."""

    __slots__ = ("binning", "data", "samples")

    def __init__(self, binning, data, samples) -> None:
        self.binning = binning
        self.data = np.asarray(data)
        if self.data.shape != (len(binning),):
            raise ValueError("unexpected shape of 'data' array")
        self.samples = np.asarray(samples)
        if self.samples.ndim != 2:
            raise ValueError("'samples' must be two-dimensional")
        if self.samples.shape[1] != len(binning):
            raise ValueError("number of bins for 'data' and 'samples' do not match")

    @property
    def num_bins(self) -> int:
        return len(self.binning)

    @property
    def num_samples(self) -> int:
        return len(self.samples)

    @property
    def covariance(self):
        return cov_from_samples(self.samples)

    @property
    def error(self):
        return np.sqrt(np.diag(self.covariance))

    @property
    def correlation(self):
        cov = self.covariance
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            std = np.sqrt(np.diag(cov))
            corr = cov / np.outer(std, std)
        corr[cov == 0] = 0
        return corr

    def __repr__(self) -> str:
        return f"{type(self).__name__}(binning={self.binning}, num_samples={self.num_samples})"

    def __eq__(self, other) -> bool:
        if not isinstance(other, type(self)):
            return NotImplemented
        return (
            self.binning == other.binning
            and np.array_equal(self.data, other.data, equal_nan=True)
            and np.array_equal(self.samples, other.samples, equal_nan=True)
        )

    def _combine(self, other, op):
        if not isinstance(other, type(self)):
            return NotImplemented
        if self.binning != other.binning or self.num_samples != other.num_samples:
            raise ValueError("binning or number of samples do not match")
        return type(self)(self.binning.copy(), op(self.data, other.data), op(self.samples, other.samples))

    def __add__(self, other):
        return self._combine(other, np.add)

    def __sub__(self, other):
        return self._combine(other, np.subtract)


class CorrData(SampledData):
    """Correlation function amplitude w(z) with jackknife samples (corrdata.py:383-)."""

    __slots__ = ()
