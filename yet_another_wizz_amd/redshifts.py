"""Clustering-redshift estimate from measured correlation functions.

Mirror of ``yaw.RedshiftData.from_corrdata`` / ``from_corrfuncs`` (src/yaw/redshifts.py:217-330):
n(z) = w_sp / sqrt(dz^2 * w_ss * w_pp), evaluated for the data and every jackknife sample. It is
the last step of the reference's end-to-end known-answer test (tests/test_setups.py:155-172), which
is why it is part of this build although it is plain post-processing on B-length vectors.
"""
from __future__ import annotations

import numpy as np

from .corrdata import CorrData, SampledData

__all__ = ["RedshiftData"]


class RedshiftData(SampledData):
    __slots__ = ()

    @classmethod
    def from_corrdata(cls, cross_data: CorrData, ref_data: CorrData | None = None, unk_data: CorrData | None = None):
        def parts(corr):
            if corr is None:
                return np.float64(1.0), np.float64(1.0)
            if corr.binning != cross_data.binning or corr.num_samples != cross_data.num_samples:
                raise ValueError("correlation data are not compatible (binning or number of samples)")
            return corr.data, corr.samples

        w_ss, w_ss_samples = parts(ref_data)
        w_pp, w_pp_samples = parts(unk_data)
        dz2 = cross_data.binning.dz ** 2
        dz2_samples = np.tile(dz2, cross_data.num_samples).reshape((cross_data.num_samples, -1))
        with np.errstate(invalid="ignore", divide="ignore"):
            data = cross_data.data / np.sqrt(dz2 * w_ss * w_pp)
            samples = cross_data.samples / np.sqrt(dz2_samples * w_ss_samples * w_pp_samples)
        return cls(cross_data.binning, data, samples)

    @classmethod
    def from_corrfuncs(cls, cross_corr, ref_corr=None, unk_corr=None):
        """Sample the correlation functions (``CorrFunc.sample()``) and combine them (redshifts.py:302-330)."""
        for corr in (ref_corr, unk_corr):
            if corr is not None:
                cross_corr.is_compatible(corr, require=True)
        return cls.from_corrdata(
            cross_corr.sample(),
            None if ref_corr is None else ref_corr.sample(),
            None if unk_corr is None else unk_corr.sample(),
        )
