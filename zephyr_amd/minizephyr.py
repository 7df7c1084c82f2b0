"""MiniZephyr / MiniZephyrHD on the GPU (interface of zephyr/backend/minizephyr.py:27-343)."""
import numpy as np
from . import _lib
from .discretization import BaseDiscretization
from .sparse import planes_to_csr


class MiniZephyr(BaseDiscretization):
    """Isotropic 2-D (visco)acoustic 9-point operator with PML; the assembly formulas of
    minizephyr.py:40-298 run in the HIP kernel `k_assemble_mz`."""

    VARIANT = _lib.HELM_MINIZEPHYR

    initMap = {
        'nPML':           (False,    '_nPML',      np.int64),
        'ky':             (False,    '_ky',        np.float64),
        'mord':           (False,    '_mord',      tuple),
    }

    @property
    def mord(self):
        'matrix ordering; only the default (+nx, +1) is supported (minizephyr.py:308-312)'
        return getattr(self, '_mord', (self.nx, +1))

    @property
    def nPML(self):
        return getattr(self, '_nPML', 10)

    @property
    def ky(self):
        return getattr(self, '_ky', 0.)

    def _assemble_args(self):
        if tuple(int(v) for v in self.mord) != (int(self.nx), 1):
            raise NotImplementedError('non-default mord re-wires the matrix (minizephyr.py:147-166); '
                                      'only (+nx,+1) is supported')
        return float(self.ky), 0.0

    @property
    def A(self):
        'The sparse system matrix, rebuilt from the device coefficient planes (minizephyr.py:300-306)'
        if getattr(self, '_A', None) is None:
            self._A = planes_to_csr(self.diagonals()[0], int(self.nz), int(self.nx))
        return self._A


class MiniZephyrHD(MiniZephyr):
    """MiniZephyr with half-differentiation of the source by default (minizephyr.py:327-343)."""

    @property
    def premul(self):
        return getattr(self, '_premul', np.sqrt(2j * np.pi * self.freq))
