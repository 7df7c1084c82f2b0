"""CPU oracle of the 3-D 27-point Helmholtz operator  ---  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

PARITY UNPINNED: the reference (uwoseis/zephyr) has no 3-D discretisation (only `ny` hooks in
zephyr/backend/base.py:20,36-40 and NotImplementedError in source.py:43-44); BASELINE config 5
asks for a 27-point operator, so this project defines one.  This file is its numpy statement;
the HIP assembly/apply/solve are checked against it, and it is itself checked against the
closed-form 3-D Green's function (the convention of zephyr/backend/analytical.py:55-59) on a
homogeneous interior window (tests/test_oracle_3d.py).

Definition (grid (nz, ny, nx), linear index (iz*ny + iy)*nx + ix, offsets o = (oz, oy, ox) in {-1,0,1}^3,
slot k = 9*(oz+1) + 3*(oy+1) + (ox+1)):

    (A u)_p = sum_o C[k(o), p] u_{p+o},      A ~ div(b grad u) + K u,   b = 1/rho, K = w~^2 / (rho c^2)

    C_o = bbar_o * ( Lx(ox) m(oy) m(oz)/dx^2 + m(ox) Ly(oy) m(oz)/dy^2 + m(ox) m(oy) Lz(oz)/dz^2 )
          + K_{p+o} * ( a * [o == 0] + (1 - a) * m(ox) m(oy) m(oz) )
    m(0) = 2/3, m(+-1) = 1/6  (trilinear-element mass weights);  a = 1/2 (lumped/consistent blend)
    bbar_o = (b_p + b_{p+o}) / 2 with edge padding
    Lx(+-1) = 1 / (xi_x(ix) * (xi_x(ix) + xi_x(ix+-1))/2),   Lx(0) = -(Lx(+1) + Lx(-1))     (same for y, z)
    xi_d(i) = 1 - i gamma_d(i) / w~,  gamma = cPML cos(pi/2 * dist / L) over nPML nodes at each end (the Eurus C-PML
    profile, zephyr/backend/eurus.py:77-97), w~ = 2 pi f - i / tau.
    Points on the boundary of the box: identity rows.
Result convention as in 2-D: u = conj(A^-1 (premul * q))  (zephyr/backend/discretization.py:101-103).
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

OFFSETS3 = [(oz, oy, ox) for oz in (-1, 0, 1) for oy in (-1, 0, 1) for ox in (-1, 0, 1)]
MASS_BLEND = 0.5


def slot3(oz, oy, ox):
    return 9 * (oz + 1) + 3 * (oy + 1) + (ox + 1)


def _m(o):
    return 2.0 / 3.0 if o == 0 else 1.0 / 6.0


def pml_profile(n, npml, h, cpml, om):
    """padded (n+2) complex stretch profile xi; gamma as eurus.py:77-91 but with index arithmetic (no arange hazard)"""
    g = np.zeros(n)
    L = h * (npml - 1)
    k = np.arange(npml)
    g[:npml] = cpml * np.cos((np.pi / 2) * (k * h / L))
    g[n - npml:] = cpml * np.cos((np.pi / 2) * ((npml - 1 - k) * h / L))
    g = np.pad(g, 1, mode='edge')
    return 1 - (1j * g) / om


def _lap_terms(xi):
    """(L_minus, L_zero, L_plus) arrays of length n from a padded profile"""
    c = xi[1:-1]
    lm = 1.0 / (c * (c + xi[:-2]) / 2)
    lp = 1.0 / (c * (c + xi[2:]) / 2)
    return lm, -(lm + lp), lp


def helm3d_coefficients(nz, ny, nx, c, rho, freq, dx=1.0, dy=None, dz=None, tau=np.inf, nPML=10, cPML=300.0):
    """C[27, nz, ny, nx] complex128"""
    dy = dx if dy is None else dy
    dz = dx if dz is None else dz
    c = np.broadcast_to(np.asarray(c, dtype=np.complex128), (nz, ny, nx)) if np.ndim(c) == 0 else np.asarray(c, np.complex128).reshape(nz, ny, nx)
    rho = np.broadcast_to(np.asarray(rho, dtype=np.float64), (nz, ny, nx)) if np.ndim(rho) == 0 else np.asarray(rho, np.float64).reshape(nz, ny, nx)
    om = 2 * np.pi * complex(freq) - 1j / tau
    Lx = _lap_terms(pml_profile(nx, nPML, dx, cPML, om))
    Ly = _lap_terms(pml_profile(ny, nPML, dy, cPML, om))
    Lz = _lap_terms(pml_profile(nz, nPML, dz, cPML, om))
    bpad = np.pad(1.0 / rho, 1, mode='edge')
    Kpad = np.pad(om * om / (rho * c ** 2), 1, mode='edge')
    b0 = bpad[1:-1, 1:-1, 1:-1]
    C = np.zeros((27, nz, ny, nx), dtype=np.complex128)
    for (oz, oy, ox) in OFFSETS3:
        bnb = bpad[1 + oz:1 + oz + nz, 1 + oy:1 + oy + ny, 1 + ox:1 + ox + nx]
        Knb = Kpad[1 + oz:1 + oz + nz, 1 + oy:1 + oy + ny, 1 + ox:1 + ox + nx]
        bbar = (b0 + bnb) / 2
        lx = Lx[ox + 1][None, None, :] * (_m(oy) * _m(oz) / dx ** 2)
        ly = Ly[oy + 1][None, :, None] * (_m(ox) * _m(oz) / dy ** 2)
        lz = Lz[oz + 1][:, None, None] * (_m(ox) * _m(oy) / dz ** 2)
        mass = (MASS_BLEND if (oz, oy, ox) == (0, 0, 0) else 0.0) + (1 - MASS_BLEND) * _m(ox) * _m(oy) * _m(oz)
        C[slot3(oz, oy, ox)] = bbar * (lx + ly + lz) + Knb * mass
    edge = np.zeros((nz, ny, nx), bool)
    edge[0], edge[-1], edge[:, 0], edge[:, -1], edge[:, :, 0], edge[:, :, -1] = True, True, True, True, True, True
    for k in range(27):
        C[k][edge] = 1.0 if k == 13 else 0.0
    return C


def coefficients_to_csr3(C):
    _, nz, ny, nx = C.shape
    N = nz * ny * nx
    iz, iy, ix = np.mgrid[0:nz, 0:ny, 0:nx]
    rows, cols, vals = [], [], []
    for k, (oz, oy, ox) in enumerate(OFFSETS3):
        jz, jy, jx = iz + oz, iy + oy, ix + ox
        ok = (jz >= 0) & (jz < nz) & (jy >= 0) & (jy < ny) & (jx >= 0) & (jx < nx)
        rows.append(((iz * ny + iy) * nx + ix)[ok]); cols.append(((jz * ny + jy) * nx + jx)[ok]); vals.append(C[k][ok])
    return sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(N, N)).tocsr()


def stencil_apply3(C, X):
    _, nz, ny, nx = C.shape
    X4 = X.reshape(nz, ny, nx, -1)
    Y = np.zeros_like(X4, dtype=np.complex128)
    for k, (oz, oy, ox) in enumerate(OFFSETS3):
        z0, z1 = max(0, -oz), nz - max(0, oz)
        y0, y1 = max(0, -oy), ny - max(0, oy)
        x0, x1 = max(0, -ox), nx - max(0, ox)
        Y[z0:z1, y0:y1, x0:x1] += C[k][z0:z1, y0:y1, x0:x1, None] * X4[z0 + oz:z1 + oz, y0 + oy:y1 + oy, x0 + ox:x1 + ox]
    return Y.reshape(X.shape)


class DirectOperator3(object):
    def __init__(self, C, premul=1.0):
        self.A = coefficients_to_csr3(C)
        self.premul = premul
        self._lu = None

    def __mul__(self, rhs):
        if self._lu is None:
            self._lu = spla.splu(self.A.tocsc())
        rhs = np.asarray(rhs, dtype=np.complex128)
        return self._lu.solve(self.premul * rhs).conjugate()


def green3d(k, r, rho=1.0, cell_volume=1.0):
    """Response of this operator to a unit entry at one node (~ delta * cell_volume):
    u = conj(A^-1 e) ~ -rho * V * exp(+i k r) / (4 pi r)  (outgoing for the conjugated convention)."""
    return -rho * cell_volume * np.exp(1j * k * r) / (4 * np.pi * r)
