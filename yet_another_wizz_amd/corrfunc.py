"""DD / DR / RD / RR bundle and the correlation estimators (mirror of
``yaw.correlation.corrfunc``, src/yaw/correlation/corrfunc.py:69-352; file I/O out of scope)."""
from __future__ import annotations

from .corrdata import CorrData
from .paircounts import NormalisedCounts

__all__ = ["CorrFunc", "EstimatorError", "davis_peebles", "landy_szalay"]


class EstimatorError(Exception):
    pass


def davis_peebles(*, dd, dr=None, rd=None, rr=None):
    """(DD - DR) / DR, preferring RD when present (corrfunc.py:69-78)."""
    if dr is None and rd is None:
        raise EstimatorError("either 'dr' or 'rd' are required")
    mixed = dr if rd is None else rd
    return (dd - mixed) / mixed


davis_peebles.name = "DP"


def landy_szalay(*, dd, dr, rd=None, rr):
    """((DD - DR) + (RR - RD)) / RR with RD defaulting to DR (corrfunc.py:81-88)."""
    if rd is None:
        rd = dr
    return ((dd - dr) + (rr - rd)) / rr


landy_szalay.name = "LS"


class CorrFunc:
    """Normalised pair counts of one correlation scale; ``sample()`` turns them into w(z)."""

    __slots__ = ("_counts_dict",)
    _kinds = ("dd", "dr", "rd", "rr")

    def __init__(self, dd, dr=None, rd=None, rr=None) -> None:
        if type(dd) is not NormalisedCounts:
            raise TypeError(f"pair counts must be of type {NormalisedCounts}")
        self._counts_dict = dict(dd=dd)
        for kind, count in (("dr", dr), ("rd", rd), ("rr", rr)):
            if count is None:
                continue
            try:
                dd.is_compatible(count, require=True)
            except ValueError as err:
                raise ValueError(f"pair counts '{kind}' and 'dd' are not compatible") from err
            self._counts_dict[kind] = count
        if len(self._counts_dict) == 1:
            raise EstimatorError("missing at least one additional pair count")

    def __repr__(self) -> str:
        kinds = "|".join(self._counts_dict)
        return (f"{type(self).__name__}(counts={kinds}, auto={self.auto}, binning={self.binning}, "
                f"num_patches={self.num_patches})")

    dd = property(lambda self: self._counts_dict["dd"])
    dr = property(lambda self: self._counts_dict.get("dr"))
    rd = property(lambda self: self._counts_dict.get("rd"))
    rr = property(lambda self: self._counts_dict.get("rr"))

    @property
    def binning(self):
        return self.dd.binning

    @property
    def auto(self) -> bool:
        return self.dd.auto

    @property
    def num_patches(self) -> int:
        return self.dd.num_patches

    @property
    def num_bins(self) -> int:
        return self.dd.num_bins

    def to_dict(self) -> dict:
        return dict(self._counts_dict)

    @classmethod
    def from_dict(cls, counts: dict):
        return cls(**counts)

    def __eq__(self, other) -> bool:
        if type(self) is not type(other):
            return NotImplemented
        mine, theirs = self.to_dict(), other.to_dict()
        return mine.keys() == theirs.keys() and all(mine[k] == theirs[k] for k in mine)

    def is_compatible(self, other, *, require: bool = False) -> bool:
        if type(self) is not type(other):
            if require:
                raise TypeError(f"{type(other)} is not compatible with {type(self)}")
            return False
        return self.dd.is_compatible(other.dd, require=require)

    def get_estimator(self):
        return davis_peebles if self.rr is None else landy_szalay

    def sample(self) -> CorrData:
        """Patch-summed counts -> estimator, for the data and each jackknife sample
        (corrfunc.py:243-272)."""
        estimator = self.get_estimator()
        values, samples = {}, {}
        for kind, counts in self._counts_dict.items():
            resampled = counts.sample_patch_sum()
            values[kind], samples[kind] = resampled.data, resampled.samples
        return CorrData(self.binning, estimator(**values), estimator(**samples))

    def bins_subset(self, item):
        return type(self).from_dict({k: c.bins_subset(item) for k, c in self._counts_dict.items()})

    def patches_subset(self, item):
        return type(self).from_dict({k: c.patches_subset(item) for k, c in self._counts_dict.items()})
