"""Redshift bin edges (mirror of ``yaw.Binning``, src/yaw/binning.py:30-160, without HDF5 I/O)."""
from __future__ import annotations

import numpy as np

from .options import Closed

__all__ = ["Binning", "parse_binning"]


def parse_binning(binning, *, optional: bool = False):
    """Validate bin edges: 1-dim, at least two, strictly increasing (binning.py:30-47)."""
    if binning is None and optional:
        return None
    edges = np.asarray(binning, dtype=np.float64)
    if edges.ndim != 1 or len(edges) < 2:
        raise ValueError("bin edges must be one-dimensionals with length > 2")
    if np.any(np.diff(edges) <= 0.0):
        raise ValueError("bin edges must increase monotonically")
    return edges


class Binning:
    __slots__ = ("edges", "closed")

    def __init__(self, edges, closed=Closed.right) -> None:
        self.edges = parse_binning(edges)
        self.closed = Closed.parse(closed)

    def __len__(self) -> int:
        return len(self.edges) - 1

    def __repr__(self) -> str:
        lb, rb = ("[", ")") if self.closed == Closed.left else ("(", "]")
        return f"{len(self)} bins @ {lb}{self.edges[0]:.3f}...{self.edges[-1]:.3f}{rb}"

    def __getstate__(self) -> dict:
        return dict(edges=self.edges, closed=self.closed)

    def __setstate__(self, state) -> None:
        for key, value in state.items():
            setattr(self, key, value)

    def __getitem__(self, item):
        left = np.atleast_1d(self.left[item])
        right = np.atleast_1d(self.right[item])
        return type(self)(np.append(left, right[-1]), closed=self.closed)

    def __iter__(self):
        for i in range(len(self)):
            yield type(self)(self.edges[i : i + 2], closed=self.closed)

    def __eq__(self, other) -> bool:
        if not isinstance(other, type(self)):
            return NotImplemented
        return np.array_equal(self.edges, other.edges) and self.closed == other.closed

    @property
    def mids(self):
        """Bin centres (binning.py:128-130); the scales are evaluated at these redshifts."""
        return (self.edges[:-1] + self.edges[1:]) / 2.0

    @property
    def left(self):
        return self.edges[:-1]

    @property
    def right(self):
        return self.edges[1:]

    @property
    def dz(self):
        return np.diff(self.edges)

    def copy(self):
        return Binning(self.edges.copy(), closed=str(self.closed))

    def assign(self, redshifts):
        """0-based bin index per object, -1 for objects outside the binning.

        Same rule as the tree builder (src/yaw/catalog/trees.py:408-414):
        ``np.digitize(z, edges, right=(closed == 'right'))`` keeps indices 1..B."""
        from ._threads import digitize

        idx = digitize(redshifts, self.edges, right=(self.closed == Closed.right))
        return np.where((idx >= 1) & (idx <= len(self)), idx - 1, -1)
