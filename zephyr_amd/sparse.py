"""Coefficient planes <-> scipy sparse matrices (the `.A` property of the operators).

Layout of the planes (see include/helm.h): C[k][iz*nx+ix], k = 3*(dz+1)+(dx+1), meaning
(A x)[iz,ix] = sum_k C[k][iz,ix] x[iz+dz, ix+dx].  The reference builds the same matrix with
scipy.sparse.diags from trimmed diagonals (zephyr/backend/minizephyr.py:147-166,252).
"""
import numpy as np
import scipy.sparse as sp


def planes_to_csr(planes, nz, nx):
    """One block of 9 planes -> (N,N) CSR; couplings that leave the grid are dropped."""
    N = nz * nx
    planes = np.asarray(planes).reshape(9, nz, nx)
    iz, ix = np.mgrid[0:nz, 0:nx]
    rows, cols, vals = [], [], []
    for k in range(9):
        dz, dx = k // 3 - 1, k % 3 - 1
        jz, jx = iz + dz, ix + dx
        inside = (jz >= 0) & (jz < nz) & (jx >= 0) & (jx < nx)
        rows.append((iz * nx + ix)[inside])
        cols.append((jz * nx + jx)[inside])
        vals.append(planes[k][inside])
    mat = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(N, N))
    return mat.tocsr()


def eurus_block_matrix(planes4, nz, nx):
    """Four blocks -> the 2N x 2N system [[M1, M2], [M3, M4]] (zephyr/backend/eurus.py:449-463)."""
    M = [planes_to_csr(planes4[m], nz, nx) for m in range(4)]
    return sp.bmat([[M[0], M[1]], [M[2], M[3]]]).tocsr()
