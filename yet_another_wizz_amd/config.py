"""Measurement configuration (mirror of ``yaw.Configuration``, src/yaw/config/classes.py:55-860).

Only what the pair-count path reads is modelled (measurements.py:99-119, :152-168):
``config.scales.{scales, num_scales, rweight, resolution}``, ``config.binning.{binning, edges,
closed, zmin, zmax, num_bins}``, ``config.cosmology`` and ``config.max_workers``.  The YAML /
paramspec machinery of the reference is out of scope; ``to_dict`` / ``from_dict`` round-trip.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .binning import Binning
from .cosmology import RedshiftBinningFactory, Scales, cosmology_is_equal, get_default_cosmology, new_scales
from .options import BinMethod, Closed, Unit

__all__ = ["Configuration", "ScalesConfig", "BinningConfig", "ConfigError"]


class ConfigError(Exception):
    pass


class _NotSet:
    def __repr__(self) -> str:
        return "NotSet"


NotSet = _NotSet()


def _updated(base: dict, **changes) -> dict:
    out = dict(base)
    out.update({k: v for k, v in changes.items() if v is not NotSet})
    return out


@dataclass(frozen=True, eq=False)
class ScalesConfig:
    scales: Scales
    rweight: float | None = None
    resolution: int | None = None

    @property
    def rmin(self):
        return self.scales.scale_min.squeeze().tolist()

    @property
    def rmax(self):
        return self.scales.scale_max.squeeze().tolist()

    @property
    def unit(self) -> str:
        return str(self.scales.unit)

    @property
    def num_scales(self) -> int:
        return self.scales.num_scales

    def __eq__(self, other) -> bool:
        if not isinstance(other, type(self)):
            return False
        return (
            np.array_equal(self.rmin, other.rmin)
            and np.array_equal(self.rmax, other.rmax)
            and self.unit == other.unit
            and self.rweight == other.rweight
            and self.resolution == other.resolution
        )

    def to_dict(self) -> dict:
        return dict(rmin=self.rmin, rmax=self.rmax, unit=self.unit, rweight=self.rweight, resolution=self.resolution)

    @classmethod
    def from_dict(cls, the_dict):
        d = dict(the_dict)
        try:
            scales = new_scales(d.pop("rmin"), d.pop("rmax"), unit=d.pop("unit", Unit.kpc))
        except Exception as err:
            raise ConfigError(str(err)) from err
        rweight, resolution = d.pop("rweight", None), d.pop("resolution", None)
        if d:
            raise ConfigError(f"unknown scales parameter(s): {sorted(d)}")
        return cls(scales, None if rweight is None else float(rweight), None if resolution is None else int(resolution))

    @classmethod
    def create(cls, *, rmin, rmax, unit=Unit.kpc, rweight=None, resolution=None):
        return cls.from_dict(dict(rmin=rmin, rmax=rmax, unit=unit, rweight=rweight, resolution=resolution))

    def modify(self, *, rmin=NotSet, rmax=NotSet, unit=NotSet, rweight=NotSet, resolution=NotSet):
        return self.from_dict(
            _updated(self.to_dict(), rmin=rmin, rmax=rmax, unit=unit, rweight=rweight, resolution=resolution)
        )


@dataclass(frozen=True, eq=False)
class BinningConfig:
    binning: Binning
    method: BinMethod = BinMethod.linear

    @property
    def edges(self):
        return self.binning.edges.tolist()

    @property
    def zmin(self) -> float:
        return float(self.binning.edges[0])

    @property
    def zmax(self) -> float:
        return float(self.binning.edges[-1])

    @property
    def num_bins(self) -> int:
        return len(self.binning)

    @property
    def closed(self) -> str:
        return str(self.binning.closed)

    @property
    def is_custom(self) -> bool:
        return self.method == BinMethod.custom

    def __eq__(self, other) -> bool:
        if not isinstance(other, type(self)):
            return False
        return self.method == other.method and self.binning == other.binning

    def to_dict(self) -> dict:
        if self.is_custom:
            return dict(method=str(self.method), edges=self.edges, closed=self.closed)
        return dict(zmin=self.zmin, zmax=self.zmax, num_bins=self.num_bins, method=str(self.method), closed=self.closed)

    @classmethod
    def from_dict(cls, the_dict, cosmology=None):
        d = dict(the_dict)
        edges, closed = d.pop("edges", None), d.pop("closed", Closed.right)
        zmin, zmax = d.pop("zmin", None), d.pop("zmax", None)
        num_bins, method = d.pop("num_bins", 30), d.pop("method", BinMethod.linear)
        if d:
            raise ConfigError(f"unknown binning parameter(s): {sorted(d)}")
        try:
            if edges is not None:
                return cls(Binning(edges, closed=closed), BinMethod.custom)
            if zmin is None or zmax is None:
                raise ConfigError("either 'edges' or 'zmin' and 'zmax' are required")
            method = BinMethod.parse(method)
            make = RedshiftBinningFactory(cosmology).get_method(method)
            return cls(make(float(zmin), float(zmax), int(num_bins), closed=closed), method)
        except ConfigError:
            raise
        except Exception as err:
            raise ConfigError(str(err)) from err

    @classmethod
    def create(cls, *, zmin=None, zmax=None, num_bins=30, method=BinMethod.linear, edges=None, closed=Closed.right,
               cosmology=None):
        return cls.from_dict(
            dict(zmin=zmin, zmax=zmax, num_bins=num_bins, method=method, edges=edges, closed=closed), cosmology
        )

    def modify(self, *, zmin=NotSet, zmax=NotSet, num_bins=NotSet, method=NotSet, edges=NotSet, closed=NotSet,
               cosmology=NotSet):
        d = _updated(self.to_dict(), zmin=zmin, zmax=zmax, num_bins=num_bins, method=method, edges=edges, closed=closed)
        if edges is NotSet and any(v is not NotSet for v in (zmin, zmax, num_bins, method)) and self.is_custom:
            d.pop("edges", None)
            d.setdefault("zmin", self.zmin)
            d.setdefault("zmax", self.zmax)
            if method is NotSet:
                d["method"] = str(BinMethod.linear)
        return self.from_dict(d, None if cosmology is NotSet else cosmology)


def _parse_cosmology(cosmology):
    if cosmology is None:
        return get_default_cosmology()
    if isinstance(cosmology, str):
        if cosmology != get_default_cosmology().name:
            raise ConfigError(f"unknown cosmology '{cosmology}' (built in: '{get_default_cosmology().name}')")
        return get_default_cosmology()
    if not (hasattr(cosmology, "comoving_distance") and hasattr(cosmology, "angular_diameter_distance")):
        raise ConfigError("'cosmology' must provide comoving_distance() and angular_diameter_distance()")
    return cosmology


@dataclass(frozen=True, eq=False)
class Configuration:
    scales: ScalesConfig
    binning: BinningConfig
    cosmology: object = None
    max_workers: int | None = None

    def __post_init__(self):
        object.__setattr__(self, "cosmology", _parse_cosmology(self.cosmology))

    def __eq__(self, other) -> bool:
        if not isinstance(other, type(self)):
            return False
        return (
            self.binning == other.binning
            and self.scales == other.scales
            and cosmology_is_equal(self.cosmology, other.cosmology)
            and self.max_workers == other.max_workers
        )

    def to_dict(self) -> dict:
        return dict(
            scales=self.scales.to_dict(),
            binning=self.binning.to_dict(),
            cosmology=getattr(self.cosmology, "name", None) or self.cosmology,
            max_workers=self.max_workers,
        )

    @classmethod
    def from_dict(cls, the_dict):
        d = dict(the_dict)
        if "scales" not in d or "binning" not in d:
            raise ConfigError("'scales' and 'binning' sections are required")
        cosmology = _parse_cosmology(d.pop("cosmology", None))
        scales = ScalesConfig.from_dict(d.pop("scales"))
        binning = BinningConfig.from_dict(d.pop("binning"), cosmology)
        max_workers = d.pop("max_workers", None)
        if d:
            raise ConfigError(f"unknown configuration parameter(s): {sorted(d)}")
        return cls(scales, binning, cosmology, max_workers)

    @classmethod
    def create(cls, *, rmin, rmax, unit=Unit.kpc, rweight=None, resolution=None, zmin=None, zmax=None, num_bins=30,
               method=BinMethod.linear, edges=None, closed=Closed.right, cosmology=None, max_workers=None):
        """Same keyword surface as the reference's ``Configuration.create`` (classes.py:689-790)."""
        return cls.from_dict(
            dict(
                scales=dict(rmin=rmin, rmax=rmax, unit=unit, rweight=rweight, resolution=resolution),
                binning=dict(zmin=zmin, zmax=zmax, num_bins=num_bins, method=method, edges=edges, closed=closed),
                cosmology=cosmology,
                max_workers=max_workers,
            )
        )

    def modify(self, *, rmin=NotSet, rmax=NotSet, unit=NotSet, rweight=NotSet, resolution=NotSet, zmin=NotSet,
               zmax=NotSet, num_bins=NotSet, method=NotSet, edges=NotSet, closed=NotSet, cosmology=NotSet,
               max_workers=NotSet):
        cosmo = self.cosmology if cosmology is NotSet else _parse_cosmology(cosmology)
        scales = self.scales.modify(rmin=rmin, rmax=rmax, unit=unit, rweight=rweight, resolution=resolution)
        binning = self.binning.modify(zmin=zmin, zmax=zmax, num_bins=num_bins, method=method, edges=edges,
                                      closed=closed, cosmology=cosmo)
        workers = self.max_workers if max_workers is NotSet else max_workers
        return type(self)(scales, binning, cosmo, workers)
