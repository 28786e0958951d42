"""Correlation scales -> angles, and the small cosmology surface that needs.

Mirrors ``yaw.cosmology`` (src/yaw/cosmology.py:94-342): ``Scales.get_angle_radian`` is the only
cosmology-dependent input of the pair-count path (B evaluations per measurement, host side).
astropy is not a dependency here: any object exposing ``angular_diameter_distance(z)`` and
``comoving_distance(z)`` in Mpc works (the reference's ``CustomCosmology`` protocol,
cosmology.py:48-91), and a self-contained flat LCDM with photons + massive neutrinos is provided
as the default (``Planck15``: parameters of Planck 2015 TT,TE,EE+lowP+lensing+ext, the reference's
default, cosmology.py:35-45).
"""
from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np

from .binning import Binning
from .options import ANGULAR_UNITS, COMOVING_UNITS, PHYSICAL_UNITS, BinMethod, Closed, Unit

__all__ = [
    "CustomCosmology",
    "FlatLCDM",
    "Planck15",
    "Scales",
    "new_scales",
    "get_default_cosmology",
    "RedshiftBinningFactory",
]

C_KM_S = 299792.458


class CustomCosmology(ABC):
    """Protocol for user supplied cosmologies (distances in Mpc)."""

    @abstractmethod
    def comoving_distance(self, z):
        ...

    @abstractmethod
    def angular_diameter_distance(self, z):
        ...


class FlatLCDM(CustomCosmology):
    """Flat LCDM with radiation: photons (Tcmb0) and neutrinos (Neff, masses in eV).

    Massive neutrinos follow the Komatsu et al. (2011, WMAP7, eq. 26) fitting function that
    astropy's FLRW models also use, so distances agree with ``astropy.cosmology.FlatLambdaCDM`` to
    quadrature accuracy (not pinned here: astropy is absent from the build image)."""

    def __init__(self, H0: float, Om0: float, *, Tcmb0: float = 0.0, Neff: float = 3.04, m_nu=(0.0, 0.0, 0.0),
                 name: str | None = None) -> None:
        self.name = name
        self.H0, self.Om0, self.Tcmb0, self.Neff = float(H0), float(Om0), float(Tcmb0), float(Neff)
        self.m_nu = np.atleast_1d(np.asarray(m_nu, dtype=np.float64))
        self.h = self.H0 / 100.0
        # photon density today: rho_gamma = 4 sigma_sb T^4 / c^3, critical density 3 H0^2 / (8 pi G)
        sigma_sb, c, G, mpc = 5.670374419e-8, 299792458.0, 6.6743e-11, 3.085677581491367e22
        rho_crit = 3.0 * (self.H0 * 1e3 / mpc) ** 2 / (8.0 * np.pi * G)
        self.Ogamma0 = 4.0 * sigma_sb * self.Tcmb0**4 / c**3 / rho_crit
        n_nu = len(self.m_nu)
        self._neff_per_nu = self.Neff / n_nu if n_nu else 0.0
        self._massive = self.m_nu[self.m_nu > 0.0]
        self._n_massless = int(np.sum(self.m_nu == 0.0))
        t_nu0 = 0.7137658555036082 * self.Tcmb0  # (4/11)^(1/3) Tcmb0
        k_b_ev = 8.617333262e-5
        self._nu_y = self._massive / (k_b_ev * t_nu0) if self.Tcmb0 > 0 and len(self._massive) else np.empty(0)
        self.Onu0 = self.Ogamma0 * self._nu_rel_density(0.0)
        self.Ode0 = 1.0 - self.Om0 - self.Ogamma0 - self.Onu0

    def _nu_rel_density(self, z):
        """Neutrino / photon density ratio at z (massive species via the Komatsu fit)."""
        prefac = 0.22710731766  # 7/8 (4/11)^(4/3)
        z = np.asarray(z, dtype=np.float64)
        if self.Tcmb0 <= 0.0 or self._neff_per_nu == 0.0:
            return np.zeros_like(z)
        if len(self._nu_y) == 0:
            return np.full_like(z, prefac * self.Neff)
        p, invp, k = 1.83, 0.54644808743, 0.3173
        y = self._nu_y / (1.0 + z[..., None])
        rel = ((1.0 + (k * y) ** p) ** invp).sum(-1) + self._n_massless
        return prefac * self._neff_per_nu * rel

    def inv_efunc(self, z):
        z = np.asarray(z, dtype=np.float64)
        zp1 = 1.0 + z
        o_rad = self.Ogamma0 * (1.0 + self._nu_rel_density(z))
        return 1.0 / np.sqrt(zp1**3 * (o_rad * zp1 + self.Om0) + self.Ode0)

    def comoving_distance(self, z):
        """Line-of-sight comoving distance in Mpc (64-point Gauss-Legendre per redshift)."""
        zs = np.atleast_1d(np.asarray(z, dtype=np.float64))
        nodes, wts = np.polynomial.legendre.leggauss(64)
        half = 0.5 * zs[:, None]
        integral = (self.inv_efunc(half * (nodes[None, :] + 1.0)) * wts[None, :]).sum(axis=1) * half[:, 0]
        out = C_KM_S / self.H0 * integral
        return out if np.ndim(z) else float(out[0])

    def angular_diameter_distance(self, z):
        return self.comoving_distance(z) / (1.0 + np.asarray(z, dtype=np.float64))

    def __repr__(self) -> str:
        return f"FlatLCDM(name={self.name!r}, H0={self.H0}, Om0={self.Om0})"


Planck15 = FlatLCDM(67.74, 0.3075, Tcmb0=2.7255, Neff=3.046, m_nu=(0.0, 0.0, 0.06), name="Planck15")


def get_default_cosmology():
    return Planck15


def cosmology_is_equal(a, b) -> bool:
    if a is b:
        return True
    if isinstance(a, FlatLCDM) and isinstance(b, FlatLCDM):
        return (a.H0, a.Om0, a.Tcmb0, a.Neff, tuple(a.m_nu)) == (b.H0, b.Om0, b.Tcmb0, b.Neff, tuple(b.m_nu))
    return False


def _as_mpc(value):
    return getattr(value, "value", value)  # tolerate astropy Quantity


class Scales:
    """Lower / upper correlation scale limits in one unit (cosmology.py:94-175)."""

    __slots__ = ("scale_min", "scale_max", "unit")

    def __init__(self, scale_min, scale_max, *, unit=Unit.kpc) -> None:
        self.unit = Unit.parse(unit)
        lo = np.atleast_1d(scale_min).astype(np.float64)
        hi = np.atleast_1d(scale_max).astype(np.float64)
        if lo.ndim != 1 or hi.ndim != 1:
            raise ValueError("min and max scales must be scalars or one-dimensional arrays")
        if len(lo) != len(hi):
            raise ValueError("number of elements in min and max scales does not match")
        if np.any((hi - lo) <= 0.0):
            raise ValueError("all min scales must be smaller than corresponding max scales")
        self.scale_min, self.scale_max = lo, hi

    def __repr__(self) -> str:
        return f"Scales(min={self.scale_min.tolist()}, max={self.scale_max.tolist()}, unit='{self.unit}')"

    @property
    def num_scales(self) -> int:
        return len(self.scale_min)

    def _to_angle(self, scales, redshift, cosmology):
        u = self.unit
        if u in ANGULAR_UNITS:  # cosmology.py:223-233
            if u == Unit.rad:
                return scales
            if u == Unit.arcsec:
                scales = scales / 3600.0
            elif u == Unit.arcmin:
                scales = scales / 60.0
            return np.deg2rad(scales)
        if u in PHYSICAL_UNITS:  # cosmology.py:250-259
            if u == Unit.kpc:
                scales = scales / 1000.0
            return scales / _as_mpc(cosmology.angular_diameter_distance(redshift))
        if u == Unit.kpc_h:  # cosmology.py:276-285
            scales = scales / 1000.0
        return scales / _as_mpc(cosmology.comoving_distance(redshift))

    def get_angle_radian(self, redshift, cosmology=None):
        """(theta_min[S], theta_max[S]) in radian at ``redshift`` (cosmology.py:158-175)."""
        cosmology = cosmology or get_default_cosmology()
        return (self._to_angle(self.scale_min, redshift, cosmology), self._to_angle(self.scale_max, redshift, cosmology))


def new_scales(scale_min, scale_max, *, unit=Unit.kpc) -> Scales:
    return Scales(scale_min, scale_max, unit=unit)


class RedshiftBinningFactory:
    """linear / comoving / logspace redshift bin edges (cosmology.py:288-342)."""

    def __init__(self, cosmology=None) -> None:
        self.cosmology = cosmology or get_default_cosmology()

    def linear(self, zmin, zmax, num_bins, *, closed=Closed.right) -> Binning:
        return Binning(np.linspace(zmin, zmax, num_bins + 1), closed=closed)

    def comoving(self, zmin, zmax, num_bins, *, closed=Closed.right) -> Binning:
        d_lo, d_hi = (_as_mpc(self.cosmology.comoving_distance(z)) for z in (zmin, zmax))
        targets = np.linspace(d_lo, d_hi, num_bins + 1)
        grid = np.linspace(zmin, zmax, 4097)  # invert the monotonic distance-redshift relation
        edges = np.interp(targets, _as_mpc(self.cosmology.comoving_distance(grid)), grid)
        edges[0], edges[-1] = zmin, zmax
        return Binning(edges, closed=closed)

    def logspace(self, zmin, zmax, num_bins, *, closed=Closed.right) -> Binning:
        lo, hi = np.log([1.0 + zmin, 1.0 + zmax])
        return Binning(np.logspace(lo, hi, num_bins + 1, base=np.e) - 1.0, closed=closed)

    def get_method(self, method=BinMethod.linear):
        method = BinMethod.parse(method)
        if method == BinMethod.custom:
            raise ValueError("custom bin edges are not generated")
        return getattr(self, str(method))
