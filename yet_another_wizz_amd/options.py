"""String options shared by the configuration and the catalogue code.

Mirrors the option vocabulary of the reference (src/yaw/options.py:39-208) without its
``strenum`` dependency: members compare equal to their plain-string value.
"""
from __future__ import annotations

from enum import Enum


class _StrOption(str, Enum):
    def __str__(self) -> str:
        return str(self.value)

    @classmethod
    def parse(cls, value):
        if isinstance(value, cls):
            return value
        try:
            return cls(str(value))
        except ValueError:
            opts = ", ".join(repr(m.value) for m in cls)
            raise ValueError(f"invalid {cls.__name__} {value!r}, expected one of: {opts}") from None


class Closed(_StrOption):
    """Which side of a redshift bin is the closed interval end (options.py:39-54)."""

    left = "left"
    right = "right"


class Unit(_StrOption):
    """Units of the correlation scales (options.py:168-208)."""

    rad = "rad"
    deg = "deg"
    arcmin = "arcmin"
    arcsec = "arcsec"
    kpc = "kpc"
    Mpc = "Mpc"
    kpc_h = "kpc/h"
    Mpc_h = "Mpc/h"


ANGULAR_UNITS = (Unit.rad, Unit.deg, Unit.arcmin, Unit.arcsec)
PHYSICAL_UNITS = (Unit.kpc, Unit.Mpc)
COMOVING_UNITS = (Unit.kpc_h, Unit.Mpc_h)


class BinMethod(_StrOption):
    linear = "linear"
    comoving = "comoving"
    logspace = "logspace"
    custom = "custom"
