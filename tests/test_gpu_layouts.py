"""GPU: buffer layouts at the C ABI.  The reference's arrays are (N, nrhs) C-order on the way in and out of `Disc * rhs`
(zephyr/backend/discretization.py:101-103; `lu.solve` returns that shape) and its sources arrive scipy-sparse
(zephyr/middleware/survey.py:162-169).  HELM_NODE_MAJOR takes exactly those arrays; helm_solve_coo takes the sparse triplets.
Every combination the direct path cannot take natively goes through transposing temporaries and must give the same numbers."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import helm_oracle as ho

pytestmark = pytest.mark.gpu


def nrm(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def model(nz=96, nx=112, seed=4):
    rng = np.random.default_rng(seed)
    c = 1900. + 900. * rng.random((nz, nx))
    rho = 1000. + 400. * rng.random((nz, nx))
    return dict(nx=nx, nz=nz, dx=10., dz=10., c=c, rho=rho, nPML=8, freq=9., rtol=1e-10)


def device_solve(op, q, layout):
    'q: (N, nrhs) host array -> (N, nrhs) through solveDevice with device buffers in the given layout'
    import torch
    dev = torch.device('cuda', op.device)
    N, nrhs = q.shape
    if layout == 'node':
        R = torch.from_numpy(np.ascontiguousarray(q)).to(dev)
        U = torch.empty((N, nrhs), dtype=torch.complex128, device=dev)
    else:
        R = torch.from_numpy(np.ascontiguousarray(q.T)).to(dev)
        U = torch.empty((nrhs, N), dtype=torch.complex128, device=dev)
    R0 = R.clone()
    op.solveDevice(R.data_ptr(), U.data_ptr(), nrhs, N, layout=layout)
    torch.cuda.synchronize()
    assert torch.equal(R, R0)                       # the right-hand sides are read-only, also where the solver uses them in place
    u = U.cpu().numpy()
    return u if layout == 'node' else u.T


def test_node_major_device_buffers_equal_rhs_major(helm_lib):
    import zephyr_amd as za
    cfg = model()
    N = cfg['nz'] * cfg['nx']
    rng = np.random.default_rng(1)
    q = np.zeros((N, 7), complex)
    q[rng.integers(0, N, 7), np.arange(7)] = rng.standard_normal(7) + 1j * rng.standard_normal(7)
    q[:, 6] += 1e-3 * (rng.standard_normal(N) + 1j * rng.standard_normal(N))          # one dense right-hand side
    for cls in (za.MiniZephyr, za.Eurus):
        op = cls(cfg)
        a = device_solve(op, q, 'rhs')
        b = device_solve(op, q, 'node')
        assert np.array_equal(a, b), (cls.__name__, nrm(b, a))     # premul = 1: the same arithmetic on the same numbers
        assert all(i['status'] == 0 and i['relres'] <= 1e-10 for i in op.lastInfo)
    # premul != 1 (half differentiation): it moves from the right-hand side to the wavefield -- same field to rounding
    op = za.MiniZephyrHD(cfg)
    a = device_solve(op, q, 'rhs')
    b = device_solve(op, q, 'node')
    assert nrm(b, a) <= 1e-12
    C = ho.minizephyr_coefficients(cfg['nz'], cfg['nx'], cfg['c'], cfg['rho'], 9., dx=10., dz=10., nPML=8)
    ref = ho.DirectOperator(C, premul=op.premul) * q
    assert nrm(b, ref) <= 1e-7


@pytest.mark.parametrize('case', ['more_than_one_batch', 'krylov', 'stacked', 'coupled_tti', 'fallback_after_failure'])
def test_node_major_where_the_direct_path_cannot_take_it(helm_lib, monkeypatch, case):
    import zephyr_amd as za
    cfg = model(56, 64)
    N = cfg['nz'] * cfg['nx']
    rng = np.random.default_rng(2)
    nrhs = 6
    rows = N
    cls = za.Eurus
    if case == 'more_than_one_batch':
        cfg['batch'] = 4
    elif case == 'krylov':
        cfg.update(method='mg', rtol=1e-9)
    elif case == 'stacked':
        rows = 2 * N
    elif case == 'coupled_tti':
        cfg.update(theta=0.3 * rng.random((cfg['nz'], cfg['nx'])), eps=0.2 * rng.random((cfg['nz'], cfg['nx'])), delta=0.1 * rng.random((cfg['nz'], cfg['nx'])), rtol=1e-8)
    elif case == 'fallback_after_failure':
        monkeypatch.setenv('HELM_TESTING', '1')
        monkeypatch.setenv('HELM_ND_INJECT_FAILURE', '1')
        cfg['rtol'] = 1e-9
    q = np.zeros((rows, nrhs), complex)
    q[rng.integers(0, rows, nrhs), np.arange(nrhs)] = 1. + 0.5j
    op = cls(cfg)
    import torch
    dev = torch.device('cuda', op.device)
    out = {}
    for layout in ('rhs', 'node'):
        R = torch.from_numpy(np.ascontiguousarray(q if layout == 'node' else q.T)).to(dev)
        U = torch.empty_like(R)
        op.solveDevice(R.data_ptr(), U.data_ptr(), nrhs, rows, layout=layout)
        torch.cuda.synchronize()
        u = U.cpu().numpy()
        out[layout] = u if layout == 'node' else u.T
    assert out['node'].shape == (rows, nrhs)
    assert nrm(out['node'], out['rhs']) <= (1e-12 if case in ('more_than_one_batch', 'stacked') else 1e-6), case


def test_operator_times_sparse_and_dense_rhs(helm_lib):
    """`Disc * q`: scipy-sparse q is expanded on the GPU (helm_solve_coo), dense q crosses as it is; both return a C-contiguous (N, nrhs)
    array and agree with the sparse LU of the oracle's matrix."""
    import zephyr_amd as za
    cfg = model(150, 170)
    N = cfg['nz'] * cfg['nx']
    locs = np.stack([np.linspace(300., 1400., 9), np.linspace(200., 900., 9)], axis=1)
    qs = za.SparseKaiserSource(cfg)(locs)
    assert sp.issparse(qs)
    for cls in (za.MiniZephyrHD, za.EurusHD):
        op = cls(cfg)
        us = op * qs
        ud = op * qs.toarray()
        assert us.shape == ud.shape == (N, 9) and us.flags['C_CONTIGUOUS'] and us.dtype == np.complex128
        assert np.array_equal(us, ud)
        u1 = op * qs.toarray()[:, 3]
        assert u1.shape == (N,) and np.array_equal(u1, ud[:, 3])
    C = ho.minizephyr_coefficients(cfg['nz'], cfg['nx'], cfg['c'], cfg['rho'], 9., dx=10., dz=10., nPML=8)
    ref = ho.DirectOperator(C, premul=za.MiniZephyrHD(cfg).premul) * qs.toarray()
    assert nrm(za.MiniZephyrHD(cfg) * qs, ref) <= 1e-7
    # the result of a large solve lives in pinned memory owned by the array (and goes back to the pool with it)
    big = za.MiniZephyr(cfg) * np.ones((N, 8), complex)
    assert big.nbytes >= (1 << 20) and big.base is not None
    with pytest.raises(ValueError):
        za.MiniZephyr(cfg) * sp.csr_matrix(qs)[:-1]
