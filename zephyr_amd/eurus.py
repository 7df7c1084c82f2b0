"""Eurus / EurusHD on the GPU (interface of zephyr/backend/eurus.py:14-552)."""
import numpy as np
from . import _lib
from .base import BaseAnisotropic
from .discretization import BaseDiscretization
from .sparse import eurus_block_matrix


class Eurus(BaseDiscretization, BaseAnisotropic):
    """TTI mixed-grid 9-point operator (Operto 2009), 2N x 2N two-field system; the assembly
    formulas of eurus.py:28-485 run in the HIP kernel `k_assemble_eurus`.

    Solves are available whenever eps == delta (isotropic and elliptical media): then M3 == 0
    and the system is block-triangular, u = M1^-1 (q1 - M2 M4^-1 q2).
    """

    VARIANT = _lib.HELM_EURUS

    initMap = {
        'nPML':           (False,    '_nPML',      np.int64),
        'freq':           (True,     None,         np.complex128),
        'mord':           (False,    '_mord',      tuple),
        'cPML':           (False,    '_cPML',      np.float64),
    }

    @property
    def mord(self):
        'matrix ordering; only the default (-nx, +1) is supported (eurus.py:494-498)'
        return getattr(self, '_mord', (-self.nx, +1))

    @property
    def cPML(self):
        return getattr(self, '_cPML', 1e3)

    @property
    def nPML(self):
        return getattr(self, '_nPML', 10)

    def _model_arrays(self):
        c, rho, _, _, _ = BaseDiscretization._model_arrays(self)
        aniso = any(getattr(self, n, None) is not None for n in ('_theta', '_eps', '_delta'))
        if not aniso:
            return c, rho, None, None, None
        dims = (int(self.nz), int(self.nx))
        return (c, rho, _lib.f64(self.theta.reshape(dims)), _lib.f64(self.eps.reshape(dims)),
                _lib.f64(self.delta.reshape(dims)))

    def _assemble_args(self):
        if tuple(int(v) for v in self.mord) != (-int(self.nx), 1):
            raise NotImplementedError('non-default mord re-wires the matrix (eurus.py:117-127); '
                                      'only (-nx,+1) is supported')
        return 0.0, float(self.cPML)

    @property
    def A(self):
        'The sparse 2N x 2N system matrix [[M1,M2],[M3,M4]] (eurus.py:449-463,487-492)'
        if getattr(self, '_A', None) is None:
            self._A = eurus_block_matrix(self.diagonals(), int(self.nz), int(self.nx))
        return self._A

    @property
    def shape(self):
        n = 2 * self.nrow
        return (n, n)

    def __mul__(self, rhs):
        'N-row rhs: zero-pad the second field and clip the result; 2N-row rhs: full result (eurus.py:512-533)'
        rhs, onedim = self._as_rhs(rhs)
        n2 = self.shape[1]
        if 2 * rhs.shape[0] == n2:
            rows = self.nrow
        elif rhs.shape[0] == n2:
            rows = n2
        else:
            raise ValueError('dimension mismatch')
        u = self._solve(rhs, rows)
        return u[:, 0] if onedim else u


class EurusHD(Eurus):
    """Eurus with half-differentiation of the source by default (eurus.py:536-552)."""

    @property
    def premul(self):
        return getattr(self, '_premul', np.sqrt(2j * np.pi * self.freq))
