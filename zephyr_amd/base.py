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

    # optional geometry keys and what they default to (base.py:31-65): a name means "the value of that attribute"
    _GEOMETRY_DEFAULTS = (('xorig', 0.), ('zorig', 0.), ('dx', 1.), ('dz', 'dx'), ('freeSurf', (False, False, False, False)))

    def __getattr__(self, name):
        # (reached only when `name` is not set: initMap stores a given value under '_' + name)
        for key, default in type(self)._GEOMETRY_DEFAULTS:
            if key == name:
                given = self.__dict__.get('_' + name)
                if given is not None:
                    return given
                return getattr(self, default) if isinstance(default, str) else default
        raise AttributeError('%s has no attribute %r' % (type(self).__name__, name))

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
