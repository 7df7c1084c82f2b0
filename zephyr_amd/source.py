"""Right-hand-side generators (interface of zephyr/backend/source.py:17-351).

Host-side input producers of the hot path: a unit delta at the nearest node, the stacked
(2N) variant used with Eurus, and Hicks' Kaiser-windowed-sinc point sources returned as a
scipy sparse (N, nsrc) matrix.  Locations are physical (x, z) pairs.
"""
import warnings
import numpy as np
import scipy.sparse as sp
from scipy.special import i0

from .base import BaseModelDependent, BaseAnisotropic


class BaseSource(BaseModelDependent):
    pass


class FakeSource(BaseSource):
    'Pass-through used with analytical systems (source.py:23-28)'

    def __call__(self, loc):
        return loc


class SimpleSource(BaseSource):
    """Unit delta at the grid node nearest to each location (source.py:31-107)."""

    def __init__(self, systemConfig):
        BaseSource.__init__(self, systemConfig)
        if hasattr(self, 'ny'):
            raise NotImplementedError('Sources not implemented for 3D case')
        nz, nx = int(self.nz), int(self.nx)
        # node coordinates exactly as np.mgrid[orig : orig + d*n : d] produces them (source.py:51-54)
        self._z, self._x = np.mgrid[self.zorig:self.zorig + self.dz * nz:self.dz,
                                    self.xorig:self.xorig + self.dx * nx:self.dx]

    def dist(self, loc):
        'distance of every node from every location, shape (nsrc, nz, nx) (source.py:56-77)'
        loc = np.asarray(loc)
        n = len(loc)
        ddx = self._x[None, :, :] - loc[:, 0].reshape((n, 1, 1))
        ddz = self._z[None, :, :] - loc[:, 1].reshape((n, 1, 1))
        return np.sqrt(ddx ** 2 + ddz ** 2)

    def linIndexOf(self, loc):
        'linear index of the nearest node (first minimum, as np.argmin) (source.py:83-88)'
        n = np.asarray(loc).shape[0]
        return np.argmin(self.dist(loc).reshape((n, -1)), axis=1)

    def vecIndexOf(self, loc):
        return self.toVecIndex(self.linIndexOf(loc))

    def __call__(self, loc):
        'dense (N, nsrc) complex right-hand sides (source.py:90-107)'
        idx = self.linIndexOf(loc)
        q = np.zeros((self.nrow, len(idx)), dtype=np.complex128)
        q[idx, np.arange(len(idx))] = 1.
        return q


class StackedSimpleSource(SimpleSource):
    """SimpleSource augmented with N zero rows for the second Eurus field (source.py:110-119)."""

    def __call__(self, loc):
        q = SimpleSource.__call__(self, loc)
        return np.vstack([q, np.zeros(q.shape, dtype=np.complex128)])


class SparseKaiserSource(SimpleSource):
    """Kaiser-windowed-sinc point sources (Hicks 2002) as a sparse (N, nsrc) matrix (source.py:122-322)."""

    initMap = {
        'ireg':           (False,    '_ireg',      np.int64),
        'freeSurf':       (False,    '_freeSurf',  tuple),
    }

    # Hicks' optimal Kaiser b for each half-width (source.py:138-149)
    HC_KAISER = {1: 1.24, 2: 2.94, 3: 4.53, 4: 6.31, 5: 7.91, 6: 9.42, 7: 10.95, 8: 12.53, 9: 14.09, 10: 14.18}

    @property
    def ireg(self):
        'half-width of the source region in nodes (default 4)'
        return getattr(self, '_ireg', 4)

    @staticmethod
    def modifyGrid(Zi, Xi, aZi, aXi):
        return Zi, Xi

    def kws(self, offset, aZi, aXi):
        'the (2 ireg + 1)^2 windowed-sinc patch for a source `offset` cells off its nearest node (source.py:156-211)'
        ireg = int(self.ireg)
        b = self.HC_KAISER.get(ireg)
        width = 2 * ireg + 1
        xoff, zoff = offset
        Zi, Xi = np.mgrid[:width, :width]
        Zi, Xi = self.modifyGrid(Zi, Xi, aZi, aXi)
        dZ = zoff + ireg - Zi
        dX = xoff + ireg - Xi
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            tZ = np.nan_to_num(np.sqrt(1 - (dZ / ireg) ** 2))
            tX = np.nan_to_num(np.sqrt(1 - (dX / ireg) ** 2))
            tZ[tZ == np.inf] = 0
            tX[tX == np.inf] = 0
        respZ = np.sinc(dZ) * (i0(b * tZ) / i0(b))
        respX = np.sinc(dX) * (i0(b * tX) / i0(b))
        return respX * respZ

    def __call__(self, sLocs):
        'sparse (N, nsrc) right-hand sides; patches are clipped (or mirrored with sign flip under a free surface) at the edges (source.py:213-317)'
        sLocs = np.asarray(sLocs)
        ireg = int(self.ireg)
        fs = self.freeSurf
        nsrc = sLocs.shape[0]
        nz, nx = int(self.nz), int(self.nx)
        N = nz * nx
        scale = 1. / (self.dx * self.dz)
        node = self.linIndexOf(sLocs)

        if ireg == 0:
            q = sp.coo_matrix((scale * np.ones(nsrc), (np.arange(nsrc), node)), shape=(nsrc, N))
            return q.T

        zsh, xsh = np.mgrid[-ireg:ireg + 1, -ireg:ireg + 1]
        lin_shift = zsh * nx + xsh
        data, rows, cols = [], [], []
        for i in range(nsrc):
            Zi, Xi = node[i] // nx, np.mod(node[i], nx)
            off = (sLocs[i][0] - self.xorig - Xi * self.dx, sLocs[i][1] - self.zorig - Zi * self.dz)
            patch = self.kws(off, Zi, Xi)
            shift = lin_shift.copy()

            if Zi < ireg:                                   # patch sticks out above iz = 0
                k = ireg - Zi
                if fs[2]:
                    lift = np.flipud(patch[:k, :])
                patch = patch[k:, :]
                shift = shift[k:, :]
                if fs[2]:
                    patch[:k, :] -= lift
            if Zi > nz - ireg - 1:                          # below iz = nz-1
                k = nz - ireg - 1 - Zi
                if fs[0]:
                    lift = np.flipud(patch[k:, :])
                patch = patch[:k, :]
                shift = shift[:k, :]
                if fs[0]:
                    patch[k:, :] -= lift
            if Xi < ireg:                                   # left of ix = 0
                k = ireg - Xi
                if fs[3]:
                    lift = np.fliplr(patch[:, :k])
                patch = patch[:, k:]
                shift = shift[:, k:]
                if fs[3]:
                    patch[:, :k] -= lift
            if Xi > nx - ireg - 1:                          # right of ix = nx-1
                k = nx - ireg - 1 - Xi
                if fs[1]:
                    lift = np.fliplr(patch[:, k:])
                patch = patch[:, :k]
                shift = shift[:, :k]
                if fs[1]:
                    patch[:, k:] -= lift

            data.append(scale * patch.ravel())
            cols.append(node[i] + shift.ravel())
            rows.append(np.full(patch.size, i))

        q = sp.coo_matrix((np.concatenate(data).astype(np.complex128), (np.concatenate(rows), np.concatenate(cols))),
                          shape=(nsrc, N), dtype=np.complex128)
        return q.T


class KaiserSource(SparseKaiserSource):
    'dense version of SparseKaiserSource (source.py:325-334)'

    def __call__(self, sLocs):
        return SparseKaiserSource.__call__(self, sLocs).toarray()


class AnisotropicKaiserSource(SparseKaiserSource, BaseAnisotropic):
    'Kaiser source on the anisotropically stretched local grid (source.py:337-351)'

    def modifyGrid(self, Zi, Xi, aZi, aXi):
        theta = self.theta[aZi, aXi]
        epsilon = self.eps[aZi, aXi]
        delta = self.delta[aZi, aXi]
        root = np.sqrt(1 + (2 * delta))
        wx = (1. + (2 * epsilon) + root) / (1 + epsilon + root)
        wz = (1. + root) / (1 + epsilon + root)
        Xn = Xi * (wx * np.cos(theta)) + Xi * (wz * np.sin(theta))
        Zn = Zi * (wx * np.sin(theta)) + Zi * (wz * np.cos(theta))
        return Zn, Xn
