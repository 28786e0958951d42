"""Angular coordinates / separations in radian and their unit-sphere (xyz, chord) counterparts.

API mirror of the reference's ``yaw.AngularCoordinates`` / ``yaw.AngularDistances``
(src/yaw/coordinates.py:72-320).  ``to_3d`` defines the float64 xyz values the device predicate
runs on, so its arithmetic follows coordinates.py:134-147 operation by operation.
"""
from __future__ import annotations

from functools import total_ordering

import numpy as np

__all__ = ["AngularCoordinates", "AngularDistances"]

TWO_PI = 2.0 * np.pi


def radec_to_xyz(ra, dec):
    """x = cos(ra) cos(dec), y = sin(ra) cos(dec), z = sin(dec)  (coordinates.py:143-146)."""
    cos_dec = np.cos(dec)
    return np.cos(ra) * cos_dec, np.sin(ra) * cos_dec, np.sin(dec)


class _ArrayBox:
    """Thin container around ``self.data`` with len / indexing / iteration / numpy interop."""

    __slots__ = ("data",)

    def __len__(self) -> int:
        return len(self.data)

    def __getitem__(self, idx):
        return type(self)(self.data[idx])

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __repr__(self) -> str:
        return f"{type(self).__name__}[{len(self)}]"

    @property
    def __array_interface__(self) -> dict:
        return self.data.__array_interface__

    def copy(self):
        return type(self)(self.data.copy())

    def tolist(self) -> list:
        return self.data.tolist()


class AngularCoordinates(_ArrayBox):
    """(N, 2) array of right ascension / declination in radian."""

    __slots__ = ()

    def __init__(self, data) -> None:
        arr = np.atleast_2d(data).astype(np.float64, copy=False)
        if arr.shape[1] != 2:
            raise ValueError("invalid coordinate dimensions, expected 2")
        self.data = arr

    @classmethod
    def from_coords(cls, coords):
        return cls(np.concatenate([np.atleast_2d(np.asarray(c)) for c in coords]))

    @classmethod
    def from_3d(cls, xyz):
        """Inverse of :meth:`to_3d` (coordinates.py:110-132); sign(0) counts as +1."""
        x, y, z = np.transpose(np.atleast_2d(xyz))
        r_xy = np.sqrt(x * x + y * y)
        r_xyz = np.sqrt(x * x + y * y + z * z)
        x_unit = np.ones_like(x)
        np.divide(x, r_xy, where=r_xy > 0.0, out=x_unit)
        sign_y = np.where(y == 0, 1.0, np.sign(y))
        ra = np.arccos(x_unit) * sign_y % TWO_PI
        dec = np.arcsin(z / r_xyz)
        return cls(np.column_stack([ra, dec]))

    @property
    def ra(self):
        return self.data[:, 0]

    @property
    def dec(self):
        return self.data[:, 1]

    def to_3d(self):
        x, y, z = radec_to_xyz(self.ra, self.dec)
        return np.column_stack([x, y, z])

    def __eq__(self, other):
        if type(self) is not type(other):
            return NotImplemented
        return self.data == other.data

    def mean(self, weights=None):
        """Weighted mean direction, averaged in xyz (coordinates.py:165-181)."""
        return type(self).from_3d(np.average(self.to_3d(), weights=weights, axis=0))

    def distance(self, other) -> "AngularDistances":
        """Great-circle separation via the chord length (coordinates.py:183-204)."""
        if not isinstance(other, type(self)):
            raise TypeError(f"cannot compute distance with type {type(other)}")
        sq = (self.to_3d() - other.to_3d()) ** 2
        return AngularDistances.from_3d(np.sqrt(sq.sum(axis=1)))


@total_ordering
class AngularDistances(_ArrayBox):
    """1-dim array of angular separations in radian."""

    __slots__ = ()

    def __init__(self, data) -> None:
        self.data = np.atleast_1d(data).astype(np.float64, copy=False)

    @classmethod
    def from_dists(cls, dists):
        return cls(np.concatenate([np.atleast_1d(np.asarray(d)) for d in dists]))

    @classmethod
    def from_3d(cls, dists):
        """Chord length on the unit sphere -> angle (coordinates.py:245-268)."""
        if np.any(np.asarray(dists) > 2.0):
            raise ValueError("distance exceeds size of unit sphere")
        return cls(2.0 * np.arcsin(np.asarray(dists) / 2.0))

    def to_3d(self):
        """Angle -> chord length r = 2 sin(theta / 2) (coordinates.py:270-277)."""
        return 2.0 * np.sin(self.data / 2.0)

    def __eq__(self, other):
        if type(self) is not type(other):
            return NotImplemented
        return self.data == other.data

    def __lt__(self, other):
        if type(self) is not type(other):
            return NotImplemented
        return self.data < other.data

    def __add__(self, other):
        if type(self) is not type(other):
            return NotImplemented
        return type(self)(self.data + other.data)

    def __sub__(self, other):
        if type(self) is not type(other):
            return NotImplemented
        return type(self)(self.data - other.data)

    def min(self):
        return type(self)(self.data.min())

    def max(self):
        return type(self)(self.data.max())
