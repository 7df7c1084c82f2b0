"""Homogeneous-medium Green's functions: the analytic pin the reference's tests use for both discretisations
(interface: zephyr/backend/analytical.py:14-80).

`green2d` / `green3d` are the free-space responses to a unit point source with the reference's sign and
scaling conventions; `AnalyticalHelmholtz` evaluates one of them on the (nz, nx) grid of a systemConfig, with
the elliptical stretch the Eurus tests use (eps, theta).  Two quirks of the reference are part of the contract
and kept: the default scale term of 0.5 (analytical.py:27), and x nodes that run up to xorig + dz * nx in
steps of dx (analytical.py:39-42), which only matters when dx != dz.
"""
import numpy as np
from scipy.special import hankel1


def green2d(k, r, rho=1., scale=0.5):
    'line source in 2-D: -i/2 H0^(1)(k r), times density and scale term; 0 where r == 0'
    r = np.asarray(r, dtype=np.float64)
    out = np.zeros(r.shape, dtype=np.complex128)
    m = r > 0
    out[m] = (-0.5j * scale * rho) * hankel1(0, k * r[m])
    return out


def green3d(k, r, rho=1., scale=0.5):
    'point source in 3-D: exp(i k r) / (4 pi r), times density and scale term; 0 where r == 0'
    r = np.asarray(r, dtype=np.float64)
    out = np.zeros(r.shape, dtype=np.complex128)
    m = r > 0
    out[m] = (scale * rho / (4. * np.pi)) * np.exp(1j * k * r[m]) / r[m]
    return out


class AnalyticalHelmholtz(object):
    """`AnalyticalHelmholtz(systemConfig)(sLocs)` -> raveled (nz * nx) complex field for the FIRST source in sLocs
    (x = first column, z = last column), like the reference class of the same name."""

    def __init__(self, systemConfig):
        sc = systemConfig
        self.c = sc['c']
        self.rho = sc.get('rho', 1.)
        self.omega = 2. * np.pi * sc['freq']
        self.k = self.omega / self.c
        self.theta = sc.get('theta', 0.)
        self.stretch = 1. / (1. + 2. * sc.get('eps', 0.))      # squared axis ratio of the elliptical wavefront
        self.scaleterm = sc.get('scaleterm', 0.5)
        self._kernel = green3d if sc.get('3D', False) else green2d
        dx, dz = sc.get('dx', 1.), sc.get('dz', 1.)
        x0, z0 = sc.get('xorig', 0.), sc.get('zorig', 0.)
        zs = np.arange(z0, z0 + dz * sc['nz'], dz)
        xs = np.arange(x0, x0 + dz * sc['nx'], dx)               # (sic) dz in the extent: analytical.py:41
        self._z, self._x = np.meshgrid(zs, xs, indexing='ij')

    def Green2D(self, r):
        return green2d(self.k, r, self.rho, self.scaleterm)

    def Green3D(self, r):
        return green3d(self.k, r, self.rho, self.scaleterm)

    def Green(self, r):
        return self._kernel(self.k, r, self.rho, self.scaleterm)

    def __call__(self, q):
        q = np.atleast_2d(q)
        ox, oz = self._x - q[0, 0], self._z - q[0, -1]
        r = np.hypot(ox, oz)
        # direction-dependent shortening of the distance: cos^2 and sin^2 have period pi, so the quadrant of the angle is immaterial
        ang = np.arctan2(oz, ox) + self.theta
        r_eff = r * np.sqrt(self.stretch * np.cos(ang) ** 2 + np.sin(ang) ** 2)
        return self.Green(r_eff).ravel()

    __mul__ = __call__
