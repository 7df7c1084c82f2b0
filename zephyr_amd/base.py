"""Grid / model geometry mix-ins (interface of zephyr/backend/base.py:11-149)."""
import numpy as np
from .config import AttributeMapper


class BaseModelDependent(AttributeMapper):
    """Grid coordinates and free-surface flags (base.py:11-109)."""

    initMap = {
        #   key            required  rename        cast
        'nx':             (True,     None,         np.int64),
        'ny':             (False,    None,         np.int64),
        'nz':             (True,     None,         np.int64),
        'xorig':          (False,    '_xorig',     np.float64),
        'zorig':          (False,    '_zorig',     np.float64),
        'dx':             (False,    '_dx',        np.float64),
        'dz':             (False,    '_dz',        np.float64),
        'freeSurf':       (False,    '_freeSurf',  tuple),
    }

    @property
    def xorig(self):
        return getattr(self, '_xorig', 0.)

    @property
    def zorig(self):
        return getattr(self, '_zorig', 0.)

    @property
    def dx(self):
        return getattr(self, '_dx', 1.)

    @property
    def dz(self):
        return getattr(self, '_dz', self.dx)

    @property
    def freeSurf(self):
        if getattr(self, '_freeSurf', None) is None:
            self._freeSurf = (False, False, False, False)
        return self._freeSurf

    @property
    def modelDims(self):
        if hasattr(self, 'ny'):
            raise NotImplementedError('3-D model geometry is not part of the 2-D operator path')
        return (self.nz, self.nx)

    @property
    def nrow(self):
        return int(np.prod(self.modelDims))

    def toLinearIndex(self, vec):
        """(n,2) array of (iz, ix) grid indices -> linear index iz*nx + ix (base.py:77-93)."""
        return vec[:, 0] * self.nx + vec[:, 1]

    def toVecIndex(self, lind):
        """linear index -> (n,2) array of (iz, ix) (base.py:95-109)."""
        return np.array([lind // self.nx, np.mod(lind, self.nx)]).T


class BaseAnisotropic(BaseModelDependent):
    """theta / eps / delta fields, zero by default (base.py:112-149)."""

    initMap = {
        'theta':          (False,    '_theta',     np.float64),
        'eps':            (False,    '_eps',       np.float64),
        'delta':          (False,    '_delta',     np.float64),
    }

    def _aniso_field(self, name):
        val = getattr(self, name, None)
        if val is None:
            val = np.zeros((self.nz, self.nx))
            setattr(self, name, val)
        if isinstance(val, np.ndarray) and val.ndim > 0:
            return val
        return val * np.ones((self.nz, self.nx), dtype=np.float64)

    @property
    def theta(self):
        return self._aniso_field('_theta')

    @property
    def eps(self):
        return self._aniso_field('_eps')

    @property
    def delta(self):
        return self._aniso_field('_delta')
