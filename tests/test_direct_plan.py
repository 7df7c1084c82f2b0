"""Elimination-tree plan of the direct solver (host side of libhelm, no GPU): the closed-form front index maps that the
HIP kernels share are exercised by running the multifrontal factorisation and solve in numpy ON the plan the library
returns, against scipy's sparse LU."""
import ctypes

import numpy as np
import pytest
import scipy.sparse.linalg as spla

from oracle import helm_oracle as ho
from zephyr_amd import _lib

OFFS = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 0), (0, 1), (1, -1), (1, 0), (1, 1)]


def plan(nz, nx, leaf):
    lib = _lib.load()
    nn = lib.helm_direct_plan(nz, nx, leaf, None, 0)
    assert nn > 0
    out = np.zeros((nn, 12), dtype=np.int32)
    assert lib.helm_direct_plan(nz, nx, leaf, out.ctypes.data_as(ctypes.c_void_p), nn) == nn
    fronts = []
    for i in range(nn):
        tot = int(out[i, 6] + out[i, 7])
        cells = np.zeros(max(tot, 1), dtype=np.int64)
        got = lib.helm_direct_plan_front(nz, nx, leaf, i, cells.ctypes.data_as(ctypes.c_void_p), tot)
        assert got == tot, 'front %d: index maps disagree (%d)' % (i, got)
        fronts.append(cells[:tot])
    return out, fronts


def multifrontal_solve(C, nz, nx, nodes, fronts, b):
    'numpy restatement of direct.hip on the library\'s plan: factor (explicit front inverses), forward, backward'
    nn = len(fronts)
    schur, fac = {}, {}
    for i in range(nn):                                   # processing order: children before parents
        s, m = int(nodes[i, 6]), int(nodes[i, 7])
        cells = fronts[i]
        loc = {int(c): a for a, c in enumerate(cells)}
        F = np.zeros((s + m, s + m), complex)
        for a, c in enumerate(cells):
            z, x = divmod(int(c), nx)
            for k, (dz, dx) in enumerate(OFFS):
                z2, x2 = z + dz, x + dx
                if not (0 <= z2 < nz and 0 <= x2 < nx):
                    continue
                bb = loc.get(z2 * nx + x2)
                if bb is None or (a >= s and bb >= s):
                    continue
                F[a, bb] = C[k][z, x]
        for kid in nodes[i, 8:10]:
            if kid >= 0:
                assert kid < i
                Sk, ck = schur.pop(int(kid))
                li = np.array([loc[int(c)] for c in ck], dtype=int)
                F[np.ix_(li, li)] += Sk
        Finv = np.linalg.inv(F[:s, :s])
        G21 = F[s:, :s] @ Finv
        F12 = F[:s, s:].copy()
        schur[i] = (F[s:, s:] - G21 @ F12, cells[s:])
        fac[i] = (Finv, G21, F12, cells[:s], cells[s:])
    assert list(schur) == [nn - 1] and schur[nn - 1][0].size == 0      # the root has no ring
    x = b.astype(complex).copy()
    for i in range(nn):
        Finv, G21, F12, S, B = fac[i]
        if B.size:
            x[B] -= G21 @ x[S]
    for i in range(nn - 1, -1, -1):
        Finv, G21, F12, S, B = fac[i]
        t = x[S] - (F12 @ x[B] if B.size else 0)
        x[S] = Finv @ t
    return x


@pytest.mark.parametrize('nz,nx,leaf', [(8, 8, 8), (9, 70, 8), (40, 50, 8), (33, 17, 4), (64, 64, 8), (50, 41, 6)])
def test_plan_reproduces_sparse_lu(nz, nx, leaf):
    rng = np.random.default_rng(nz * 1000 + nx)
    c = 2000. + 1500. * rng.random((nz, nx))
    C = ho.minizephyr_coefficients(nz, nx, c, ho.gardner_rho(c.astype(complex)), 12.0, dx=10., dz=10., nPML=5).reshape(9, nz, nx)
    A = ho.coefficients_to_csr(C)
    nodes, fronts = plan(nz, nx, leaf)
    # every cell is eliminated exactly once
    seps = np.concatenate([f[:int(nodes[i, 6])] for i, f in enumerate(fronts)])
    assert np.array_equal(np.sort(seps), np.arange(nz * nx))
    # padded sizes cover the group
    assert np.all(nodes[:, 6] <= nodes[:, 10]) and np.all(nodes[:, 7] <= nodes[:, 11])
    b = rng.standard_normal(nz * nx) + 1j * rng.standard_normal(nz * nx)
    x = multifrontal_solve(C, nz, nx, nodes, fronts, b)
    xr = spla.splu(A.tocsc()).solve(b)
    assert np.linalg.norm(x - xr) / np.linalg.norm(xr) < 1e-9
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-9


def test_plan_shape_at_bench_size():
    nodes, = (plan_nodes_only(1024, 1024, 8),)
    assert nodes.shape[0] > 30000
    root = nodes[-1]
    assert root[6] == 1024 and root[7] == 0 and root[4] in (0, 1)
    leaves = nodes[nodes[:, 4] < 0]
    assert leaves[:, 6].max() <= 64 and leaves[:, 7].max() <= 4 * 8 + 4


def plan_nodes_only(nz, nx, leaf):
    lib = _lib.load()
    nn = lib.helm_direct_plan(nz, nx, leaf, None, 0)
    out = np.zeros((nn, 12), dtype=np.int32)
    lib.helm_direct_plan(nz, nx, leaf, out.ctypes.data_as(ctypes.c_void_p), nn)
    return out
