"""Closed-form homogeneous Green's functions used as a test oracle by the reference's own tests
(interface of zephyr/backend/analytical.py:14-80), including its conventions: hankel1 with a
0.5 scale term, and the x-grid built with dz as the extent step (analytical.py:39-42)."""
import warnings
import numpy as np
from scipy.special import hankel1


class AnalyticalHelmholtz(object):

    def __init__(self, systemConfig):
        self.omega = 2 * np.pi * systemConfig['freq']
        self.c = systemConfig['c']
        self.rho = systemConfig.get('rho', 1.)
        self.k = self.omega / self.c
        self.stretch = 1. / (1 + (2. * systemConfig.get('eps', 0.)))
        self.theta = systemConfig.get('theta', 0.)
        self.scaleterm = systemConfig.get('scaleterm', 0.5)

        xorig = systemConfig.get('xorig', 0.)
        zorig = systemConfig.get('zorig', 0.)
        dx = systemConfig.get('dx', 1.)
        dz = systemConfig.get('dz', 1.)
        nx = systemConfig['nx']
        nz = systemConfig['nz']
        # NB: the x extent uses dz, as in the reference (analytical.py:41)
        self._z, self._x = np.mgrid[zorig:zorig + dz * nz:dz, xorig:xorig + dz * nx:dx]
        self.Green = self.Green3D if systemConfig.get('3D', False) else self.Green2D

    def Green2D(self, r):
        return self.scaleterm * self.rho * (-0.5j * hankel1(0, self.k * r))

    def Green3D(self, r):
        return self.scaleterm * self.rho * (1. / (4 * np.pi * r)) * np.exp(1j * self.k * r)

    def __call__(self, q):
        x = q[0, 0]
        z = q[0, -1]
        ddx = self._x - x
        ddz = self._z - z
        dist = np.sqrt(ddx ** 2 + ddz ** 2)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            angle = np.arctan(ddz / ddx) + self.theta
            stretch = np.sqrt(self.stretch * np.cos(angle) ** 2 + np.sin(angle) ** 2)
            return np.nan_to_num(self.Green(dist * stretch)).ravel()

    def __mul__(self, q):
        return self(q)
