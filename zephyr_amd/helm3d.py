"""3-D 27-point Helmholtz operator on the GPU (BASELINE config 5).

The reference has no 3-D discretisation (zephyr/backend/base.py:20,36-40 reserves `ny`/`dy`;
zephyr/backend/source.py:43-44 raises NotImplementedError), so there is no class to mirror: `Helm3D`
follows the same `systemConfig` contract as the 2-D discretisations (keys nx, ny, nz, dx, dy, dz, c, rho,
freq, tau, premul, nPML, cPML) and the same result convention conj(A^-1 (premul rhs)).  The operator is
defined in oracle/helm3d_oracle.py (trilinear-element-weighted 27-point star, C-PML on all six faces).
Solves: BiCGSTAB right-preconditioned by the layer-preserving 3-D multigrid of libhelm (mg3d.hip) whose coarsest level is solved directly;
Jacobi-preconditioned BiCGSTAB / CGNR on request.
"""
import ctypes
import numpy as np
import scipy.sparse as sp

from . import _lib
from .discretization import BaseDiscretization


class Helm3D(BaseDiscretization):

    heavyPrepare = True          # the prepare step builds the multigrid preconditioner (helm_prefactor_n): dispatchers keep strictly one item ahead

    VARIANT = _lib.HELM_3D

    initMap = {
        #   key            required  rename        cast
        'ny':             (True,     None,         np.int64),
        'dy':             (False,    '_dy',        np.float64),
        'nPML':           (False,    '_nPML',      np.int64),
        'cPML':           (False,    '_cPML',      np.float64),
    }

    @property
    def dy(self):
        return getattr(self, '_dy', self.dx)

    @property
    def nPML(self):
        return getattr(self, '_nPML', 10)

    @property
    def cPML(self):
        return getattr(self, '_cPML', 300.)

    @property
    def modelDims(self):
        return (int(self.nz), int(self.ny), int(self.nx))

    @property
    def c(self):
        if isinstance(self._c, np.ndarray) and self._c.ndim > 0:
            return self._c
        return self._c * np.ones(self.modelDims, dtype=np.complex128)

    @property
    def rho(self):
        if hasattr(self, '_rho'):
            if not (isinstance(self._rho, np.ndarray) and self._rho.ndim > 0):
                return self._rho * np.ones(self.modelDims, dtype=np.float64)
        else:
            self._rho = 310. * self.c.real ** 0.25
        return self._rho

    @property
    def method(self):
        return getattr(self, '_method', 'bicgstab')

    @property
    def handle(self):
        if getattr(self, '_handle', None) is None:
            lib = _lib.load()
            _lib.require_gpu()
            nz, ny, nx = self.modelDims
            h = lib.helm_create3d(self.device, nz, ny, nx, float(self.dx), float(self.dy), float(self.dz), int(self.nPML))
            if not h:
                raise _lib.HelmError(-2, _lib.last_error(None))
            try:
                c = _lib.c128(self.c.reshape(self.modelDims))
                rho = _lib.f64(self.rho.reshape(self.modelDims))
                _lib.check(lib.helm_set_model(h, _lib.ptr(c), _lib.ptr(rho), None, None, None), h)
                f = complex(self.freq)
                _lib.check(lib.helm_assemble(h, f.real, f.imag, float(self.tau), 0.0, float(self.cPML)), h)
            except Exception:
                lib.helm_destroy(h)
                raise
            self._handle = h
        return self._handle

    def diagonals(self):
        'coefficient planes from the device: (27, nz, ny, nx) complex128'
        lib = _lib.load()
        out = np.empty((27,) + self.modelDims, dtype=np.complex128)
        _lib.check(lib.helm_get_diagonals(self.handle, _lib.ptr(out)), self.handle)
        return out

    @property
    def A(self):
        if getattr(self, '_A', None) is None:
            C = self.diagonals()
            nz, ny, nx = self.modelDims
            N = nz * ny * nx
            iz, iy, ix = np.mgrid[0:nz, 0:ny, 0:nx]
            rows, cols, vals = [], [], []
            for k in range(27):
                oz, oy, ox = k // 9 - 1, (k // 3) % 3 - 1, k % 3 - 1
                jz, jy, jx = iz + oz, iy + oy, ix + ox
                ok = (jz >= 0) & (jz < nz) & (jy >= 0) & (jy < ny) & (jx >= 0) & (jx < nx)
                rows.append(((iz * ny + iy) * nx + ix)[ok]); cols.append(((jz * ny + jy) * nx + jx)[ok]); vals.append(C[k][ok])
            self._A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(N, N)).tocsr()
        return self._A
