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


def test_host_arrays_go_through_pinned_chunks_or_straight(helm_lib):
    """Round 6: a host array that crosses the C ABI (helm_apply, helm_get_diagonals, helm_set_model) is copied through the library's pinned 4-MB chunks unless
    it is pinned memory already (helm_host_alloc), which goes straight -- the caller's pageable pages are never registered with the runtime (include/helm.h).
    Both ways move the same bytes: sizes that end inside a chunk, on a chunk boundary and below one chunk, up and down, against the oracle's matrix."""
    import zephyr_amd as za
    from zephyr_amd import _lib
    cfg = model(300, 273)
    op = za.MiniZephyr(cfg)
    N = cfg['nz'] * cfg['nx']
    planes = op.diagonals()[0]
    rng = np.random.default_rng(8)
    for nrhs in (1, 3, 7):                                     # 1.3 MB (one chunk, partly filled), 3.9 MB, 9.2 MB (three chunks, the last one partial)
        x = rng.standard_normal((N, nrhs)) + 1j * rng.standard_normal((N, nrhs))
        y_pageable = op.applyForward(x)
        xp = _lib.pinned_empty((nrhs, N)); xp[...] = x.T
        yp = _lib.pinned_empty((nrhs, N))
        _lib.check(helm_lib.helm_apply(op.handle, 0, 0, _lib.ptr(xp), _lib.ptr(yp), nrhs), op.handle)
        assert np.array_equal(y_pageable, yp.T)
        assert nrm(y_pageable, ho.stencil_apply(planes, x)) <= 1e-12
    cfgb = model(512, 512)
    opb = za.MiniZephyr(cfgb)
    xb = rng.standard_normal((512 * 512, 2)) + 1j * rng.standard_normal((512 * 512, 2))       # 8 MB: two full chunks, nothing left over
    yb = opb.applyForward(xb)
    xp = _lib.pinned_empty((2, 512 * 512)); xp[...] = xb.T
    yp = _lib.pinned_empty((2, 512 * 512))
    _lib.check(helm_lib.helm_apply(opb.handle, 0, 0, _lib.ptr(xp), _lib.ptr(yp), 2), opb.handle)
    assert np.array_equal(yb, yp.T)
    d1, d2 = opb.diagonals(), opb.diagonals()                  # 37.7 MB down through the chunks, twice
    assert np.array_equal(d1, d2) and np.all(np.isfinite(d1.view(float)))


@pytest.mark.parametrize('cls_name,premul', [('Eurus', None), ('MiniZephyr', None), ('Eurus', 0.3 - 1.7j), ('MiniZephyrHD', None), ('MiniZephyr', 0.3 - 1.7j)])
def test_direct_output_writes_the_wavefields_of_the_two_step_path(helm_lib, monkeypatch, cls_name, premul):
    """Round 5 (helm_tuning.nd_direct_out): for a full-width node-major batch the back substitution writes u = conj(premul x) into the caller's array
    itself -- leaf cells there only, separator cells there and in the scratch the levels below read -- and the residual launch reads the caller's
    array instead of storing it.  Bit for bit the wavefields of the path that stores them from the residual launch (HELM_ND_DIRECT_OUT=0), with point
    sources, a dense column and an all-zero one, sparse-rhs skipping on and off; the reported residual is the same number up to the rounding of u."""
    import zephyr_amd as za
    nz, nx, nrhs = 150, 170, 200
    rng = np.random.default_rng(11)
    c = 1800. + 2000. * rng.random((nz, nx))
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, freq=8., nPML=8, rtol=1e-10, method='direct', batch=256)
    if premul is not None:
        cfg['premul'] = premul
    locs = np.stack([rng.uniform(100., 10. * nx - 100., nrhs), rng.uniform(20., 60., nrhs)], axis=1)
    q = za.SparseKaiserSource(cfg)(locs).toarray()
    q[:, 3] = 0.0
    q[:, -1] = rng.standard_normal(nz * nx) + 1j * rng.standard_normal(nz * nx)
    q[:, 5] *= (0.2 + 0.9j)                                    # a complex source term
    monkeypatch.setenv('HELM_ND_POISON', '1')
    for sparse in ('1', '0'):
        monkeypatch.setenv('HELM_ND_SPARSE_RHS', sparse)
        out, info = {}, {}
        for mode in ('1', '0'):
            monkeypatch.setenv('HELM_ND_DIRECT_OUT', mode)
            op = getattr(za, cls_name)(cfg)
            out[mode] = device_solve(op, q, 'node')
            info[mode] = [dict(i) for i in op.lastInfo]
            del op.factors
        if not np.array_equal(out['1'], out['0']):             # (diagnostics: which cells / columns differ, and by how much)
            bad = np.argwhere(out['1'] != out['0'])
            where = [(int(c // nx), int(c % nx), int(j), complex(out['1'][c, j]), complex(out['0'][c, j])) for c, j in bad[:8]]
            cols = sorted(set(int(j) for _, j in bad))[:12]
            its = {m: sorted(set(i['iterations'] for i in info[m])) for m in info}
            rr = {m: (min(i['relres'] for i in info[m]), max(i['relres'] for i in info[m])) for m in info}
            raise AssertionError('direct output differs from the two-step path (sparse=%s): %d of %d entries, relative %.2e; passes %s; relres range %s; columns %s; first (z, x, col, direct, two-step): %s'
                                 % (sparse, len(bad), out['1'].size, nrm(out['1'], out['0']), its, rr, cols, where[:2]))
        assert not np.any(out['1'][:, 3])
        for a, b in zip(info['1'], info['0']):
            assert a['status'] == b['status'] == 0 and a['iterations'] == b['iterations'] == 1
            # the residual launch reads conj(premul x) rounded to the caller's array where the two-step path had x itself (premul != 1, which the HD
            # classes also have): residuals of a few 1e-15 then differ in their leading digits, larger ones agree
            assert abs(a['relres'] - b['relres']) <= 1e-3 * b['relres'] + 1e-14
    op = getattr(za, cls_name)(cfg)
    C = ho.minizephyr_coefficients(nz, nx, op.c, op.rho, complex(op.freq), dx=10., dz=10., nPML=8) if cls_name.startswith('Mini') else None
    ref = (ho.DirectOperator(C, premul=op.premul) if C is not None else
           ho.DirectOperator(ho.eurus_coefficients(nz, nx, c, op.rho, 8., dx=10., dz=10., nPML=8), eurus=True, premul=op.premul)) * q[:, [0, 5, nrhs - 1]]
    assert nrm(out['1'][:, [0, 5, nrhs - 1]], ref) <= 1e-7


def test_direct_output_hands_x_back_when_a_refinement_pass_is_needed(helm_lib, monkeypatch):
    """With the pivoted-LU treatment of ill-conditioned fronts off, the 512^2 model at 16 Hz needs a second pass: the first residual check reads the
    caller's array (direct output), x is rebuilt from it for the refinement pass, and the final wavefields are those of the two-step path to rounding
    (premul != 1: the rebuilt x differs from the original in its last bits, which the refinement pass then removes to rtol)."""
    import zephyr_amd as za
    from zephyr_amd.models import marmousi_like
    n, nrhs, dx = 512, 160, 9.0
    c = marmousi_like(n, n, dx)
    cfg = dict(nx=n, nz=n, dx=dx, dz=dx, c=c, freq=16., nPML=10, cPML=1e3, rtol=1e-12, method='direct', batch=256, premul=0.8 + 0.3j)
    locs = np.stack([np.linspace(300., dx * n - 300., nrhs), np.full(nrhs, 20.)], axis=1)
    q = za.SparseKaiserSource(cfg)(locs).toarray()
    q[(n // 2) * n + n // 3, 1] = 1j; q[40 * n + 400, 2] = 1. - 1j          # two sources inside the model as well
    monkeypatch.setenv('HELM_ND_STABLE', '0')
    out, passes = {}, {}
    for mode in ('1', '0'):
        monkeypatch.setenv('HELM_ND_DIRECT_OUT', mode)
        op = za.Eurus(cfg)
        out[mode] = device_solve(op, q, 'node')
        passes[mode] = max(i['iterations'] for i in op.lastInfo)
        assert all(i['status'] in (0, 3) and i['relres'] <= 1e-10 for i in op.lastInfo), op.lastInfo[:3]
        del op.factors
    assert passes['1'] >= 2 and passes['0'] >= 2, passes             # (the case does exercise the hand-back)
    assert nrm(out['1'], out['0']) <= 1e-9
