"""GPU: the sparse direct path (direct.hip) -- dense kernels against numpy, wavefields against sparse LU and the
reference's golden vectors, factor re-use, refinement and error conventions."""
import ctypes
import os

import numpy as np
import pytest

from oracle import helm_oracle as ho

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def nrm(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def crand(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


@pytest.mark.parametrize('M,N,K,batch', [(64, 64, 8, 1), (36, 256, 64, 7), (1, 1, 1, 3), (65, 67, 9, 2), (130, 33, 71, 3), (9, 256, 9, 40),
                                         # tall-and-skinny products (<= 16 columns, long inner dimension): the streaming kernel of the 3-D coarse solve
                                         (100, 16, 500, 3), (8, 16, 384, 1), (893, 16, 2900, 2), (37, 5, 1000, 2), (64, 1, 2000, 1), (13, 9, 447, 4),
                                         # ... and few enough of them that the inner dimension is split over workgroups (chunks that do not divide it, an empty last chunk)
                                         (300, 16, 1027, 1), (1900, 16, 3001, 2), (129, 7, 1024, 5), (40, 16, 1032, 9),
                                         # 49-row fronts (the leaves): three blocks of 16 rows on the matrix cores + one row on the vector ALUs
                                         (49, 256, 81, 5), (49, 64, 49, 3), (49, 100, 7, 2), (49, 300, 83, 2), (49, 32, 49, 4)])
def test_batched_zgemm(helm_lib, M, N, K, batch):
    rng = np.random.default_rng(M * 7 + N)
    A, B, C = crand(rng, batch, M, K), crand(rng, batch, K, N), crand(rng, batch, M, N)
    for alpha, beta in ((1 + 0j, 0j), (-1 + 0j, 1 + 0j), (0.3 - 0.2j, 0.5 + 0.1j)):
        ref = alpha * (A @ B) + (beta * C if beta != 0 else 0)
        out = np.ascontiguousarray(C if beta != 0 else np.full_like(C, np.nan))     # beta == 0 must not read C
        al, be = np.array([alpha.real, alpha.imag]), np.array([beta.real, beta.imag])
        rc = helm_lib.helm_debug_zgemm(0, M, N, K, al.ctypes.data_as(ctypes.c_void_p), np.ascontiguousarray(A).ctypes.data_as(ctypes.c_void_p),
                                       np.ascontiguousarray(B).ctypes.data_as(ctypes.c_void_p), be.ctypes.data_as(ctypes.c_void_p),
                                       out.ctypes.data_as(ctypes.c_void_p), batch)
        assert rc == 0
        assert np.abs(out - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()) * K


@pytest.mark.parametrize('n,batch', [(1, 2), (5, 3), (32, 4), (33, 2), (64, 5), (100, 2), (257, 1),
                                     (32, 2100), (8, 2048), (20, 2300), (64, 2050)])     # >= 2048 blocks: the wave-per-matrix kernel
def test_batched_inverse(helm_lib, n, batch):
    rng = np.random.default_rng(n)
    A = crand(rng, batch, n, n) + 2.0 * np.sqrt(n) * np.eye(n)        # leading blocks comfortably invertible
    if n == 5 or batch > 2000:
        A[0, 0, 0] = 0.0                                              # forces a row exchange in the base block
        if batch > 2000 and n >= 8:
            A[-1, :4, :4] = np.fliplr(np.eye(4)) * 3.0               # and a permutation-like leading block in the last matrix
    out = np.ascontiguousarray(A.copy())
    assert helm_lib.helm_debug_inverse(0, n, out.ctypes.data_as(ctypes.c_void_p), batch) == 0
    for b in range(batch):
        assert np.abs(out[b] @ A[b] - np.eye(n)).max() <= 1e-10


@pytest.mark.parametrize('n,batch', [(512, 2), (545, 1), (1024, 1), (1101, 3), (40, 7), (64, 40), (100, 3), (129, 300), (257, 2)])
def test_one_launch_block_step_agrees_with_the_two_launch_form(helm_lib, n, batch, monkeypatch):
    """From 512 to 1536 unknowns a block step of the Gauss-Jordan inversion is ONE launch that reads one copy of the matrix and writes the other
    (k_gj_step, direct.hip); HELM_ND_GJSTEP=0 is the panel copy + update pair.  Same pivots, the products in another order: the inverses agree
    to rounding -- even and odd numbers of steps (the odd ones end in the workspace and are copied back), a last block of 1 and of 13 columns."""
    rng = np.random.default_rng(7 * n)
    A = crand(rng, batch, n, n) + 2.0 * np.sqrt(n) * np.eye(n)
    A[0, 0, 0] = 0.0
    A[-1, 32:36, 32:36] = np.fliplr(np.eye(4)) * 3.0
    monkeypatch.setenv('HELM_ND_GJSTEP_MIN', '33')                     # (default 512: below that the two-launch form is kept; more matrices than tiles
    outs = []                                                          #  per matrix put the sweeps in several z-slices of the grid)
    for flag in ('0', '1'):
        monkeypatch.setenv('HELM_ND_GJSTEP', flag)
        out = np.ascontiguousarray(A.copy())
        assert helm_lib.helm_debug_inverse(0, n, out.ctypes.data_as(ctypes.c_void_p), batch) == 0
        outs.append(out)
    for b in range(batch):
        assert np.abs(outs[1][b] @ A[b] - np.eye(n)).max() <= 1e-9
        assert np.abs(outs[1][b] - outs[0][b]).max() <= 1e-11 * np.abs(outs[0][b]).max() * n


@pytest.mark.parametrize('n,batch', [(512, 1), (600, 2), (1000, 1), (1101, 3), (3100, 1)])
def test_large_inverse_with_look_ahead_and_block_recursion(helm_lib, n, batch):
    """From 512 unknowns up the blocked Gauss-Jordan sweeps the next pivot block inside the launch of the current rank-32 update; from 3000
    up one level of the 2 x 2 block recursion comes first.  Sizes that are not multiples of 32 / 64, several matrices per launch, row
    exchanges inside pivot blocks."""
    rng = np.random.default_rng(n)
    A = crand(rng, batch, n, n) + 2.0 * np.sqrt(n) * np.eye(n)
    A[0, 0, 0] = 0.0
    A[-1, 32:36, 32:36] = np.fliplr(np.eye(4)) * 3.0                   # a permutation-like piece in the second pivot block
    out = np.ascontiguousarray(A.copy())
    assert helm_lib.helm_debug_inverse(0, n, out.ctypes.data_as(ctypes.c_void_p), batch) == 0
    for b in range(batch):
        assert np.abs(out[b] @ A[b] - np.eye(n)).max() <= 1e-9


@pytest.mark.parametrize('cls', ['MiniZephyr', 'Eurus'])
@pytest.mark.parametrize('nz,nx', [(64, 64), (70, 90), (41, 150), (9, 9)])
def test_direct_matches_sparse_lu(helm_lib, cls, nz, nx):
    import zephyr_amd as za
    rng = np.random.default_rng(nz + nx)
    c = 1800. + 2200. * rng.random((nz, nx))
    rho = 1000. + 600. * rng.random((nz, nx))
    npml = 8 if min(nz, nx) > 20 else 2
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, rho=rho, freq=9., nPML=npml, rtol=1e-11, method='direct')
    src = np.stack([np.linspace(20., 10. * nx - 30., 5), np.linspace(15., 10. * nz - 25., 5)], 1)
    q = za.SimpleSource(cfg)(src)
    op = getattr(za, cls)(cfg)
    u = op * q
    if cls == 'MiniZephyr':
        ref = ho.DirectOperator(ho.minizephyr_coefficients(nz, nx, c, rho, 9., dx=10., dz=10., nPML=npml)) * q
    else:
        ref = ho.DirectOperator(ho.eurus_coefficients(nz, nx, c, rho, 9., dx=10., dz=10., nPML=npml), eurus=True) * q
    assert nrm(u, ref) <= 1e-9, op.lastInfo
    assert all(i['status'] == 0 and i['relres'] <= 1e-11 and i['method'] == 4 and 1 <= i['iterations'] <= 4 for i in op.lastInfo)
    t = op.lastTiming()
    assert t['factor_ms'] > 0
    u2 = op * q                              # factors are kept on the handle
    assert op.lastTiming()['factor_ms'] == 0
    assert np.array_equal(u, u2)             # and the path is bit-reproducible


def test_direct_golden_wavefields_and_stacked_eurus(helm_lib):
    import zephyr_amd as za
    g = np.load(os.path.join(GOLD, 'g9_xhlayr.npz'))
    nz, nx = g['c'].shape
    sc = dict(nx=nx, nz=nz, dx=1., dz=1., c=g['c'], freq=float(g['freq']), method='direct', rtol=1e-11)
    op = za.MiniZephyrHD(sc)
    u = op * za.SparseKaiserSource(sc)(g['src'])
    R = za.SparseKaiserSource(sc)(g['rec']).T
    assert nrm(R * u, g['data']) <= 1e-8
    # Eurus, 2N-row stacked right-hand side: block-triangular solve with two factorisations (M4 then M1)
    nz, nx = 48, 56
    rng = np.random.default_rng(5)
    c = 2000. + 1000. * rng.random((nz, nx))
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, freq=8., nPML=6, method='direct', rtol=1e-11)
    q = np.zeros((2 * nz * nx, 2), complex)
    q[20 * nx + 30, 0] = 1.0; q[nz * nx + 25 * nx + 12, 1] = 1.0; q[nz * nx + 10 * nx + 40, 0] = 0.5
    op = za.Eurus(cfg)
    u = op * q
    ref = ho.DirectOperator(ho.eurus_coefficients(nz, nx, c, ho.gardner_rho(c.astype(complex)), 8., dx=10., dz=10., nPML=6), eurus=True) * q
    assert nrm(u, ref) <= 1e-8


def test_direct_new_frequency_refactors_and_unsupported_cases(helm_lib):
    import zephyr_amd as za
    nz, nx = 40, 44
    c = np.full((nz, nx), 2500.)
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, freq=6., nPML=6, method='direct')
    q = za.SimpleSource(cfg)(np.array([[200., 210.]]))
    dist = za.MultiFreq(dict(cfg, Disc=za.MiniZephyr, freqs=[6., 11.]))
    us = list(dist * q)
    for f, u in zip((6., 11.), us):
        ref = ho.DirectOperator(ho.minizephyr_coefficients(nz, nx, c, ho.gardner_rho(c.astype(complex)), f, dx=10., dz=10., nPML=6)) * q
        assert nrm(u, ref) <= 1e-9
    # 3-D operators have no direct path
    with pytest.raises(Exception):
        za.Helm3D(dict(nx=12, ny=12, nz=12, dx=10., c=2000., freq=5., method='direct')) * np.ones(12 ** 3, complex)


@pytest.mark.parametrize('nz,nx', [(44, 52), (96, 120)])
def test_direct_coupled_tti_system(helm_lib, nz, nx):
    """eps != delta: the 2N x 2N system [[M1, M2], [M3, M4]] (eurus.py:430-464) factored with two unknowns per cell; N-row and
    stacked 2N-row right-hand sides against the sparse LU of the reference-identical matrix"""
    import zephyr_amd as za
    rng = np.random.default_rng(nz)
    c = 1800. + 2200. * rng.random((nz, nx))
    rho = 1000. + 600. * rng.random((nz, nx))
    theta = 0.5 * rng.random((nz, nx)) - 0.25
    eps = 0.25 * rng.random((nz, nx))
    delta = 0.12 * rng.random((nz, nx))
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, rho=rho, freq=9., nPML=6, theta=theta, eps=eps, delta=delta, rtol=1e-10)
    C4 = ho.eurus_coefficients(nz, nx, c, rho, 9., dx=10., dz=10., nPML=6, theta=theta, eps=eps, delta=delta)
    assert np.abs(C4[2]).max() > 0
    lu = ho.DirectOperator(C4, eurus=True)
    q = za.SimpleSource(cfg)(np.array([[250., 220.], [330., 150.], [120., 300.]]))
    for method in ('direct', 'auto'):
        op = za.Eurus(dict(cfg, method=method))
        u = op * q
        assert nrm(u, lu * q) <= 1e-8, op.lastInfo
        assert all(i['method'] == 4 and i['relres'] <= 1e-10 and i['iterations'] <= 4 for i in op.lastInfo), op.lastInfo
        q2 = np.vstack([q, 0.3j * q[::-1]])
        u2 = op * q2
        assert u2.shape == q2.shape and nrm(u2, lu * q2) <= 1e-8, op.lastInfo


def test_auto_prefers_direct_and_falls_back_to_krylov(helm_lib, monkeypatch):
    import zephyr_amd as za
    nz, nx = 64, 72
    rng = np.random.default_rng(11)
    c = 2000. + 1500. * rng.random((nz, nx))
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, freq=7., nPML=8, rtol=1e-10)
    q = za.SimpleSource(cfg)(np.array([[300., 320.], [500., 200.]]))
    ref = ho.DirectOperator(ho.eurus_coefficients(nz, nx, c, ho.gardner_rho(c.astype(complex)), 7., dx=10., dz=10., nPML=8), eurus=True) * q
    op = za.Eurus(cfg)                       # method defaults to 'auto'
    u = op * q
    assert all(i['method'] == 4 for i in op.lastInfo) and nrm(u, ref) <= 1e-8
    monkeypatch.setenv('HELM_TESTING', '1')                 # the fault-injection hooks are inert without it
    monkeypatch.setenv('HELM_ND_INJECT_FAILURE', '1')
    op2 = za.Eurus(cfg)
    u2 = op2 * q
    assert all(i['method'] == 3 and i['status'] == 0 and i['iterations'] > 5 for i in op2.lastInfo), op2.lastInfo
    assert nrm(u2, ref) <= 1e-7
    with pytest.raises(Exception) as ei:
        za.Eurus(dict(cfg, method='direct')) * q
    assert 'injected' in str(ei.value)
    monkeypatch.delenv('HELM_ND_INJECT_FAILURE')
    assert helm_lib.helm_trim() == 0


def test_auto_partial_fallback_resolves_only_the_stalled_sources(helm_lib, monkeypatch):
    """when a few right-hand sides miss rtol on the direct path, only those are handed to the Krylov path (the
    others keep the direct result) and the factors are released at once."""
    import zephyr_amd as za
    nz, nx = 64, 72
    rng = np.random.default_rng(12)
    c = 2000. + 1500. * rng.random((nz, nx))
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, freq=7., nPML=8, rtol=1e-10)
    q = za.SimpleSource(cfg)(np.array([[300., 320.], [500., 200.], [120., 400.], [610., 510.], [333., 111.]]))
    ref = ho.DirectOperator(ho.eurus_coefficients(nz, nx, c, ho.gardner_rho(c.astype(complex)), 7., dx=10., dz=10., nPML=8), eurus=True) * q
    monkeypatch.setenv('HELM_TESTING', '1')
    monkeypatch.setenv('HELM_ND_INJECT_STALL', '2')
    for cls, r in ((za.Eurus, ref), (za.MiniZephyr, None)):
        op = cls(cfg)
        u = op * q
        assert [i['method'] for i in op.lastInfo] == [3, 3, 4, 4, 4], op.lastInfo
        assert all(i['status'] == 0 and i['relres'] <= 1e-10 for i in op.lastInfo)
        assert op.factors is False or True          # property stays readable
        if r is not None:
            assert nrm(u, r) <= 1e-7
        else:
            rr = op.applyForward(u.conj()) - q
            assert np.linalg.norm(rr, axis=0).max() <= 2e-10 * np.linalg.norm(q, axis=0).min()
    monkeypatch.delenv('HELM_ND_INJECT_STALL')


def test_direct_free_surface_and_viscous_configurations(helm_lib):
    """configurations the reference supports on the MiniZephyr side: free surfaces (sign-flipped identity rows), complex
    velocity (Q), Laplace damping tau, dx != dz"""
    import zephyr_amd as za
    nz, nx = 60, 84
    rng = np.random.default_rng(2)
    c = (2200. + 800. * rng.random((nz, nx))) * (1 + 0.5j / 80.)
    rho = 1800. + 300. * rng.random((nz, nx))
    cfg = dict(nx=nx, nz=nz, dx=8., dz=12., c=c, rho=rho, freq=10., nPML=7, tau=0.6, freeSurf=(True, False, False, True),
               method='direct', rtol=1e-11)
    q = za.SimpleSource(cfg)(np.array([[250., 300.], [420., 90.]]))
    op = za.MiniZephyr(cfg)
    u = op * q
    C = ho.minizephyr_coefficients(nz, nx, c, rho, 10., dx=8., dz=12., nPML=7, tau=0.6, freeSurf=(True, False, False, True))
    assert nrm(u, ho.DirectOperator(C) * q) <= 1e-9


def test_sparse_rhs_upload_and_receiver_sampling(helm_lib):
    'helm_rhs_from_coo_device / helm_sample_device against numpy'
    import torch
    import scipy.sparse as sp
    import zephyr_amd as za
    nz, nx = 30, 34
    N = nz * nx
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=2500., freq=6., nPML=5)
    op = za.MiniZephyr(cfg)
    q = za.SparseKaiserSource(cfg)(np.array([[55., 61.], [203., 148.], [12., 250.]]))       # (N, 3) sparse, clipped at the edge
    dev = torch.device('cuda', op.device)
    R = torch.full((3, N), float('nan'), dtype=torch.complex128, device=dev)
    op.rhsFromSparseDevice(q, R.data_ptr())
    assert np.array_equal(R.cpu().numpy(), q.toarray().T)
    rng = np.random.default_rng(4)
    U = rng.standard_normal((3, N)) + 1j * rng.standard_normal((3, N))
    Rm = sp.csr_matrix((za.SparseKaiserSource(cfg)(np.array([[40., 40.], [100., 120.], [250., 33.], [170., 222.]])) * (0.5 + 0.25j)).T)
    csr = (torch.from_numpy(Rm.indptr.astype(np.int64)).to(dev), torch.from_numpy(Rm.indices.astype(np.int64)).to(dev),
           torch.from_numpy(Rm.data.astype(np.complex128)).to(dev), 4)
    dU = torch.from_numpy(U).to(dev)
    out = torch.empty((4, 3), dtype=torch.complex128, device=dev)
    op.sampleDevice(dU.data_ptr(), 3, csr, out.data_ptr())
    ref = Rm @ U.T
    assert np.abs(out.cpu().numpy() - ref).max() <= 1e-13 * np.abs(ref).max()


@pytest.mark.parametrize('nz,nx,nrhs', [(3, 3, 1), (3, 200, 2), (200, 3, 1), (5, 7, 7), (17, 33, 3), (129, 65, 1)])
def test_direct_degenerate_shapes(helm_lib, nz, nx, nrhs):
    'grids thinner than a leaf, single cells of interior, odd sizes, 1..7 right-hand sides'
    import zephyr_amd as za
    rng = np.random.default_rng(nz * 31 + nx)
    c = 2000. + 1000. * rng.random((nz, nx))
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, freq=7. + 0.3j, nPML=2, method='direct', rtol=1e-11, premul=0.5 - 2j)
    q = rng.standard_normal((nz * nx, nrhs)) + 1j * rng.standard_normal((nz * nx, nrhs))
    op = za.MiniZephyr(cfg)
    u = op * q
    C = ho.minizephyr_coefficients(nz, nx, c, ho.gardner_rho(c.astype(complex)), 7. + 0.3j, dx=10., dz=10., nPML=2)
    ref = ho.DirectOperator(C, premul=0.5 - 2j) * q
    assert nrm(u, ref) <= 1e-9, op.lastInfo
    u1 = op * q[:, 0]                       # 1-D right-hand side keeps its shape
    assert u1.shape == (nz * nx,) and nrm(u1, ref[:, 0]) <= 1e-9


@pytest.mark.parametrize('seed', range(40))
def test_direct_randomised_configurations(helm_lib, seed):
    """seeded sweep over grid shapes, spacings, frequencies (incl. complex), PML widths, free surfaces, damping, ky, Q, anisotropy
    and right-hand-side counts: every combination goes through the direct path and is compared with the sparse LU of the oracle"""
    import zephyr_amd as za
    rng = np.random.default_rng(1000 + seed)
    nz, nx = int(rng.integers(12, 90)), int(rng.integers(12, 90))
    dx, dz = float(rng.choice([5., 10., 12.5])), float(rng.choice([5., 10., 8.]))
    npml = int(rng.integers(2, max(3, min(nz, nx) // 3)))
    f = float(rng.uniform(3., 25.)) + (1j * float(rng.uniform(0., 0.5)) if rng.random() < 0.3 else 0.)
    c = 1500. + 2500. * rng.random((nz, nx))
    if rng.random() < 0.4:
        c = c * (1 + 0.5j / rng.uniform(30., 200.))
    rho = 1000. + 800. * rng.random((nz, nx))
    nrhs = int(rng.integers(1, 9))
    q = rng.standard_normal((nz * nx, nrhs)) + 1j * rng.standard_normal((nz * nx, nrhs))
    tau = float(rng.uniform(0.3, 3.)) if rng.random() < 0.3 else np.inf
    cfg = dict(nx=nx, nz=nz, dx=dx, dz=dz, c=c, rho=rho, freq=f, nPML=npml, tau=tau, method='direct', rtol=1e-10)
    if rng.random() < 0.5:
        fs = tuple(bool(b) for b in rng.integers(0, 2, 4))
        ky = float(rng.uniform(0., 0.002)) if rng.random() < 0.3 else 0.
        cfg.update(freeSurf=fs, ky=ky)
        op = za.MiniZephyr(cfg)
        C = ho.minizephyr_coefficients(nz, nx, c, rho, f, dx=dx, dz=dz, nPML=npml, tau=tau, ky=ky, freeSurf=fs)
        ref = ho.DirectOperator(C) * q
    else:
        kw = {}
        if rng.random() < 0.5:          # elliptical anisotropy with tilt: eps == delta keeps the system block-triangular
            e = 0.25 * rng.random((nz, nx))
            kw = dict(eps=e, delta=e, theta=0.6 * rng.random((nz, nx)) - 0.3)
        cpml = float(rng.choice([1e3, 3e2]))
        try:
            C4 = ho.eurus_coefficients(nz, nx, c, rho, f, dx=dx, dz=dz, nPML=npml, tau=tau, cPML=cpml, **kw)
        except ValueError:
            pytest.skip('the reference raises for this PML length (np.arange hazard)')
        cfg.update(cPML=cpml, **kw)
        op = za.Eurus(cfg)
        ref = ho.DirectOperator(C4, eurus=True) * q
    u = op * q
    assert nrm(u, ref) <= 1e-7, (seed, op.lastInfo)
    assert all(i['status'] == 0 and i['method'] == 4 for i in op.lastInfo)


@pytest.mark.parametrize('seed', range(8))
def test_direct_randomised_coupled_tti(helm_lib, seed):
    'seeded sweep of genuinely coupled TTI systems (eps != delta) through the two-unknowns-per-cell direct path'
    import zephyr_amd as za
    rng = np.random.default_rng(5000 + seed)
    nz, nx = int(rng.integers(16, 70)), int(rng.integers(16, 70))
    npml = int(rng.integers(3, 7))
    f = float(rng.uniform(5., 14.))
    c = 1800. + 2000. * rng.random((nz, nx))
    rho = 1000. + 500. * rng.random((nz, nx))
    theta = 0.8 * rng.random((nz, nx)) - 0.4
    eps = 0.3 * rng.random((nz, nx))
    delta = 0.15 * rng.random((nz, nx))
    nrhs = int(rng.integers(1, 5))
    rows = nz * nx * (2 if seed % 2 else 1)
    q = rng.standard_normal((rows, nrhs)) + 1j * rng.standard_normal((rows, nrhs))
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, rho=rho, freq=f, nPML=npml, theta=theta, eps=eps, delta=delta, rtol=1e-9)
    try:
        C4 = ho.eurus_coefficients(nz, nx, c, rho, f, dx=10., dz=10., nPML=npml, theta=theta, eps=eps, delta=delta)
    except ValueError:
        pytest.skip('the reference raises for this PML length (np.arange hazard)')
    op = za.Eurus(cfg)
    u = op * q
    assert nrm(u, ho.DirectOperator(C4, eurus=True) * q) <= 1e-6, (seed, op.lastInfo)
    assert all(i['method'] == 4 and i['status'] == 0 for i in op.lastInfo), op.lastInfo


def test_direct_many_right_hand_sides_in_batches(helm_lib):
    'more right-hand sides than one batch holds: 300 sources with batch=128 go through three passes of the same factors'
    import zephyr_amd as za
    nz, nx = 48, 40
    rng = np.random.default_rng(8)
    c = 2000. + 1500. * rng.random((nz, nx))
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, freq=9., nPML=6, method='direct', rtol=1e-11, batch=128)
    q = rng.standard_normal((nz * nx, 300)) + 1j * rng.standard_normal((nz * nx, 300))
    op = za.MiniZephyr(cfg)
    u = op * q
    ref = ho.DirectOperator(ho.minizephyr_coefficients(nz, nx, c, ho.gardner_rho(c.astype(complex)), 9., dx=10., dz=10., nPML=6)) * q
    assert nrm(u, ref) <= 1e-9
    assert len(op.lastInfo) == 300 and all(i['status'] == 0 for i in op.lastInfo)
    assert op.lastTiming()['factor_ms'] > 0            # one factorisation for all three batches


def test_unreachable_rtol_reports_fp64_floor(helm_lib):
    """rtol below what fp64 can represent: refinement stalls, the library evaluates the floor eps (|| |A||x| || + ||q||) / ||q|| of the true
    residual and reports the right-hand sides that sit on it as status 3 (solved to the precision the arithmetic has), not as failures."""
    import zephyr_amd as za
    nz, nx = 96, 120
    rng = np.random.default_rng(21)
    c = 1800. + 2200. * rng.random((nz, nx))
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, freq=9., nPML=8, rtol=1e-17, method='direct')
    q = za.SimpleSource(cfg)(np.array([[300., 320.], [800., 500.]]))
    op = za.Eurus(cfg)
    u = op * q                                   # does not raise
    assert [i['status'] for i in op.lastInfo] == [3, 3], op.lastInfo
    assert all(1e-17 < i['relres'] < 1e-12 for i in op.lastInfo), op.lastInfo
    ref = ho.DirectOperator(ho.eurus_coefficients(nz, nx, c, ho.gardner_rho(c.astype(complex)), 9., dx=10., dz=10., nPML=8), eurus=True) * q
    assert nrm(u, ref) <= 1e-9


def test_ill_conditioned_fronts_reeliminated_with_lu(helm_lib, monkeypatch):
    """direct.hip, NdStable (default; HELM_ND_STABLE=0 switches it off): fronts whose pivot block is near-singular -- subdomains close to a resonance at this
    frequency -- are eliminated again with one pivoted LU used for the Schur complement, the forward and the backward pass.  On the
    512^2 bench model at 16 Hz four of 8191 fronts are taken (condition numbers 1e4 ... 9e5) and the first pass then meets rtol 1e-10
    where the explicit inverses alone need a refinement pass (first-pass residual 1.2e-9); same wavefield either way, and the same as
    the sparse LU of the oracle's matrix."""
    import zephyr_amd as za
    from zephyr_amd.models import marmousi_like
    from oracle import helm_oracle as ho
    n, dx, f = 512, 9.0, 16.0
    c = marmousi_like(n, n, dx)
    cfg = dict(nx=n, nz=n, dx=dx, dz=dx, c=c, nPML=10, cPML=1e3, freq=f, rtol=1e-10, method='direct')
    q = np.zeros((n * n, 3), complex)
    q[2 * n + n // 5, 0] = 1.; q[(n // 2) * n + n // 3, 1] = 1j; q[40 * n + 400, 2] = 1. - 1j
    monkeypatch.setenv('HELM_ND_STABLE', '0')
    op0 = za.Eurus(cfg)
    u0 = op0 * q
    passes0 = max(i['iterations'] for i in op0.lastInfo)
    monkeypatch.delenv('HELM_ND_STABLE')
    op1 = za.Eurus(cfg)
    u1 = op1 * q
    assert all(i['status'] == 0 and i['relres'] <= 1e-10 for i in op1.lastInfo), op1.lastInfo
    passes1 = max(i['iterations'] for i in op1.lastInfo)
    assert passes1 == 1 and passes0 == 2, (passes0, passes1)
    assert np.linalg.norm(u1 - u0) / np.linalg.norm(u0) <= 1e-8
    u1b = op1 * q                                   # factors (and the treated fronts) are re-used
    assert np.array_equal(u1, u1b)
    rho = ho.gardner_rho(c)
    C4 = ho.eurus_coefficients(n, n, c, rho, f, dx=dx, dz=dx, nPML=10, cPML=1e3)
    ref = ho.DirectOperator(C4[0]) * q
    assert np.linalg.norm(u1 - ref) / np.linalg.norm(ref) <= 1e-7
    del op0.factors, op1.factors


@pytest.mark.parametrize('seed', range(3))
def test_lu_treatment_forced_on_many_fronts(helm_lib, monkeypatch, seed):
    """The same machinery with the threshold lowered until dozens of fronts on every watched level are taken (HELM_ND_STABLE_THR=30): rebuild of
    the front from its children, pivoted LU, Schur complement through triangular solves, forward / backward fix-ups -- against the sparse LU
    of the oracle's matrix, MiniZephyr with free surfaces and Eurus, odd grid shapes, several right-hand sides."""
    import zephyr_amd as za
    rng = np.random.default_rng(77 + seed)
    nz, nx = int(rng.integers(150, 260)), int(rng.integers(150, 300))
    c = 1600. + 2200. * rng.random((nz, nx))
    rho = 1000. + 600. * rng.random((nz, nx))
    f = float(rng.uniform(8., 20.))
    nrhs = int(rng.integers(3, 40))
    q = np.zeros((nz * nx, nrhs), complex)
    q[rng.integers(0, nz * nx, nrhs), np.arange(nrhs)] = rng.standard_normal(nrhs) + 1j * rng.standard_normal(nrhs)
    monkeypatch.setenv('HELM_ND_STABLE_THR', '30')
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, rho=rho, freq=f, nPML=8, method='direct', rtol=1e-10)
    if seed % 2 == 0:
        fs = (True, False, False, False)
        op = za.MiniZephyr(dict(cfg, freeSurf=fs))
        C = ho.minizephyr_coefficients(nz, nx, c, rho, f, dx=10., dz=10., nPML=8, freeSurf=fs)
        ref = ho.DirectOperator(C) * q
    else:
        op = za.Eurus(cfg)
        C4 = ho.eurus_coefficients(nz, nx, c, rho, f, dx=10., dz=10., nPML=8)
        ref = ho.DirectOperator(C4[0]) * q
    u = op * q
    assert all(i['status'] == 0 and i['relres'] <= 1e-10 for i in op.lastInfo), op.lastInfo
    assert nrm(u, ref) <= 1e-7, (seed, nrm(u, ref))
    del op.factors


@pytest.mark.parametrize('force_pivoted', [0, 1])
@pytest.mark.parametrize('cls,nz,nx,fs', [('MiniZephyr', 64, 64, (0, 0, 0, 0)), ('Eurus', 70, 90, (0, 0, 0, 0)), ('MiniZephyr', 41, 150, (1, 0, 0, 1)),
                                          ('Eurus', 128, 128, (0, 0, 0, 0)), ('MiniZephyr', 57, 33, (0, 1, 1, 0))])
def test_fused_leaf_level_matches_sparse_lu(helm_lib, monkeypatch, cls, nz, nx, fs, force_pivoted):
    """The leaf level in one kernel (k_leaf_factor: banded LU + per-column substitution straight from the coefficient planes) on grids whose
    leaves have every shape between 3 and 8 cells a side, with free surfaces, and -- force_pivoted -- every leaf re-done by the row-pivoted
    fall-back kernel.  (By default only leaf groups of 2048 fronts or more take that kernel: the small grids of this suite would never see it.)"""
    import zephyr_amd as za
    monkeypatch.setenv('HELM_ND_FUSEDLEAF_MIN', '1')
    if force_pivoted:
        monkeypatch.setenv('HELM_LEAF_DBG', '8')
    rng = np.random.default_rng(nz * 3 + nx)
    c = 1700. + 2500. * rng.random((nz, nx))
    rho = 1000. + 500. * rng.random((nz, nx))
    cfg = dict(nx=nx, nz=nz, dx=10., dz=12., c=c, rho=rho, freq=11., nPML=6, rtol=1e-11, method='direct', freeSurf=fs)
    op = getattr(za, cls)(cfg)
    q = za.SimpleSource(cfg)(np.array([[nx * 3.3, nz * 4.1], [nx * 6.0, nz * 7.7], [25., 30.]]))
    u = op * q
    if cls == 'MiniZephyr':
        C = ho.minizephyr_coefficients(nz, nx, c, rho, 11., dx=10., dz=12., nPML=6, freeSurf=fs)
        ref = ho.DirectOperator(C) * q
    else:
        ref = ho.DirectOperator(ho.eurus_coefficients(nz, nx, c, rho, 11., dx=10., dz=12., nPML=6), eurus=True) * q
    assert nrm(u, ref) <= 1e-8
    assert all(i['status'] == 0 and i['relres'] <= 1e-11 for i in op.lastInfo), op.lastInfo
    # ... and bit-for-bit what the batched leaf path returns is NOT required (other elimination order), only the same wavefield
    monkeypatch.setenv('HELM_ND_FUSEDLEAF', '0')
    op2 = getattr(za, cls)(cfg)
    assert nrm(op2 * q, u) <= 1e-9


def test_fused_leaf_level_at_coarse_sampling_falls_back_where_it_must(helm_lib, monkeypatch):
    """Few points per wavelength make the leaf blocks indefinite: elimination without pivoting then meets small pivots in some of them, the
    kernel flags those leaves and the pivoted kernel re-does them -- the wavefield must come out right either way."""
    import zephyr_amd as za
    monkeypatch.setenv('HELM_ND_FUSEDLEAF_MIN', '1')
    nz = nx = 96
    rng = np.random.default_rng(5)
    c = 1500. + 300. * rng.random((nz, nx))
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, freq=30., nPML=8, rtol=1e-10, method='direct')          # ~5 points per wavelength
    op = za.MiniZephyr(cfg)
    q = za.SimpleSource(cfg)(np.array([[300., 320.], [610., 450.]]))
    u = op * q
    C = ho.minizephyr_coefficients(nz, nx, c, op.rho, 30., dx=10., dz=10., nPML=8)
    ref = ho.DirectOperator(C) * q
    assert nrm(u, ref) <= 1e-7
    assert all(i['status'] in (0, 3) for i in op.lastInfo), op.lastInfo



@pytest.mark.parametrize('nsrc', [5, 130])
def test_declared_support_gives_the_wavefields_of_the_scan(helm_lib, monkeypatch, nsrc):
    """helm_set_rhs_support: the support of the sparse source matrix, made by the library from its triplets, stands in for the scan of the dense
    right-hand sides at the leaf level of the forward pass.  Device-resident solve in the reference's (N, nsrc) layout: bit for bit the wavefields of
    the solve that looks for the nonzeros itself, with the front-vector arena poisoned; HELM_ND_SUPPORT_CHECK=1 passes on the true support and fails the
    solve when a source is missing from the declared one; the declaration is forgotten after one solve."""
    import torch
    import scipy.sparse as sp
    import zephyr_amd as za
    nz, nx = 150, 170
    N = nz * nx
    rng = np.random.default_rng(50 + nsrc)
    c = 1800. + 2000. * rng.random((nz, nx))
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, freq=8., nPML=8, rtol=1e-10, method='direct', batch=256)
    locs = np.stack([rng.uniform(100., 10. * nx - 100., nsrc), rng.uniform(20., 60., nsrc)], axis=1)
    qs = sp.csc_matrix(za.SparseKaiserSource(cfg)(locs))
    dev = torch.device('cuda', 0)
    monkeypatch.setenv('HELM_ND_POISON', '1')
    op = za.Eurus(cfg)
    R = torch.zeros((N, nsrc), dtype=torch.complex128, device=dev)
    op.rhsFromSparseDevice(qs, R.data_ptr(), layout='node')
    bits = op.rhsSupportFromSparse(qs)
    assert int(bits.count_nonzero()) == len(np.unique(qs.tocoo().row))
    U0 = torch.empty_like(R); U1 = torch.empty_like(R); U2 = torch.empty_like(R)
    op.solveDevice(R.data_ptr(), U0.data_ptr(), nsrc, N, layout='node')                       # the library scans
    monkeypatch.setenv('HELM_ND_SUPPORT_CHECK', '1')
    op.solveDevice(R.data_ptr(), U1.data_ptr(), nsrc, N, layout='node', support=bits)         # declared (and verified)
    monkeypatch.delenv('HELM_ND_SUPPORT_CHECK')
    op.solveDevice(R.data_ptr(), U2.data_ptr(), nsrc, N, layout='node')                       # one shot: this one scans again
    torch.cuda.synchronize()
    assert torch.equal(torch.view_as_real(U0), torch.view_as_real(U1))
    assert torch.equal(torch.view_as_real(U0), torch.view_as_real(U2))
    assert bool(torch.isfinite(torch.view_as_real(U1)).all())
    # a support that misses the first source: the check refuses the solve
    wrong_full = torch.zeros_like(bits)
    coo = qs[:, 1:].tocoo()
    wrong_full[torch.from_numpy(np.unique(coo.row)).to(dev)] = 0xFF
    lone = np.setdiff1d(np.unique(qs[:, [0]].tocoo().row), np.unique(coo.row))
    if lone.size:                                                    # (the first source has rows no other source touches)
        monkeypatch.setenv('HELM_ND_SUPPORT_CHECK', '1')
        with pytest.raises(Exception):
            op.solveDevice(R.data_ptr(), U1.data_ptr(), nsrc, N, layout='node', support=wrong_full)
    del op.factors

@pytest.mark.parametrize('nsrc', [5, 130])
def test_declared_support_may_be_a_superset_of_the_nonzeros(helm_lib, monkeypatch, nsrc):
    """a bit of helm_set_rhs_support means the block MAY be nonzero, so supersets are legal -- every bit set on point sources (a
    conservative caller), and a scipy matrix with explicit zero entries (muted source terms: helm_rhs_support_from_coo sets bits from the
    triplets without looking at the values).  A leaf flagged without a nonzero must store its zeros: with the front-vector arena poisoned, rows
    a parent reads on the word of a raised flag would otherwise be NaN.  Bit for bit the wavefields of the scanning solve."""
    import torch
    import scipy.sparse as sp
    import zephyr_amd as za
    nz, nx = 150, 170
    N = nz * nx
    rng = np.random.default_rng(150 + nsrc)
    c = 1800. + 2000. * rng.random((nz, nx))
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, freq=8., nPML=8, rtol=1e-10, method='direct', batch=256)
    locs = np.stack([rng.uniform(100., 10. * nx - 100., nsrc), rng.uniform(20., 60., nsrc)], axis=1)
    qs = sp.csc_matrix(za.SparseKaiserSource(cfg)(locs))
    dev = torch.device('cuda', 0)
    monkeypatch.setenv('HELM_ND_POISON', '1')
    op = za.Eurus(cfg)
    R = torch.zeros((N, nsrc), dtype=torch.complex128, device=dev)
    op.rhsFromSparseDevice(qs, R.data_ptr(), layout='node')
    U0 = torch.empty_like(R); U1 = torch.empty_like(R); U2 = torch.empty_like(R)
    op.solveDevice(R.data_ptr(), U0.data_ptr(), nsrc, N, layout='node')                       # the library scans
    exact = op.rhsSupportFromSparse(qs)
    everything = torch.full_like(exact, (1 << ((nsrc + 63) // 64)) - 1)
    monkeypatch.setenv('HELM_ND_SUPPORT_CHECK', '1')
    op.solveDevice(R.data_ptr(), U1.data_ptr(), nsrc, N, layout='node', support=everything)
    # explicit zeros: entries scattered over the grid (far from the sources too) whose value is 0.0
    coo = qs.tocoo()
    zr = rng.integers(0, N, 40 * nsrc); zc = rng.integers(0, nsrc, 40 * nsrc)
    qz = sp.coo_matrix((np.concatenate([coo.data, np.zeros(zr.size, dtype=coo.data.dtype)]),
                        (np.concatenate([coo.row, zr]), np.concatenate([coo.col, zc]))), shape=qs.shape)
    bits_z = op.rhsSupportFromSparse(qz)
    assert int(bits_z.count_nonzero()) > int(exact.count_nonzero())
    op.solveDevice(R.data_ptr(), U2.data_ptr(), nsrc, N, layout='node', support=bits_z)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(torch.view_as_real(U1)).all()) and bool(torch.isfinite(torch.view_as_real(U2)).all())
    assert torch.equal(torch.view_as_real(U0), torch.view_as_real(U1))
    assert torch.equal(torch.view_as_real(U0), torch.view_as_real(U2))
    assert all(i['status'] == 0 for i in op.lastInfo), op.lastInfo
    del op.factors


@pytest.mark.parametrize('nsrc', [3, 70, 200])
def test_sparse_right_hand_sides_skip_nothing_that_matters(helm_lib, monkeypatch, nsrc):
    """The forward pass leaves out the fronts whose right-hand-side rows and whose children's rows are all zero in a block of 64 columns
    (GemmRows::act).  Point sources, sources in some column blocks only, a dense random right-hand side and an all-zero column: bit for bit the
    wavefields of the pass that computes every front (HELM_ND_SPARSE_RHS=0)."""
    import zephyr_amd as za
    nz, nx = 150, 170
    rng = np.random.default_rng(nsrc)
    c = 1800. + 2000. * rng.random((nz, nx))
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, freq=8., nPML=8, rtol=1e-10, method='direct', batch=256)
    locs = np.stack([rng.uniform(100., 10. * nx - 100., nsrc), rng.uniform(20., 60., nsrc)], axis=1)       # near the surface, like a survey's sources
    q = za.SparseKaiserSource(cfg)(locs).toarray()
    q[:, 1] = 0.0                                                    # a column without a source
    if nsrc > 64:
        q[:, 64:128] = 0.0                                           # a whole block of 64 columns without any
        q[:, -1] = rng.standard_normal(nz * nx) + 1j * rng.standard_normal(nz * nx)      # and a dense one
    out = {}
    monkeypatch.setenv('HELM_ND_POISON', '1')              # unwritten front-vector rows are NaN: a read that a flag should have masked cannot hide
    for mode in ('1', '0'):
        monkeypatch.setenv('HELM_ND_SPARSE_RHS', mode)
        op = za.Eurus(cfg)
        out[mode] = op * q
        assert all(i['status'] == 0 for j, i in enumerate(op.lastInfo) if np.any(q[:, j])), op.lastInfo
        del op.factors
    assert np.array_equal(out['1'], out['0'])
    assert not np.any(out['1'][:, 1])
    ref = ho.DirectOperator(ho.eurus_coefficients(nz, nx, c, za.Eurus(cfg).rho, 8., dx=10., dz=10., nPML=8), eurus=True) * q[:, [0, 2, nsrc - 1]]
    assert nrm(out['1'][:, [0, 2, nsrc - 1]], ref) <= 1e-7


def test_short_product_kernels_and_listed_levels_change_no_bit(helm_lib, monkeypatch):
    """Round 5 (helm_tuning.nd_leaf_idle).  On sparse right-hand sides three things leave the tile kernel's slab pipeline: the leaf back substitution of
    the (leaf, 64-column block) pairs without a right-hand side (k_leaf_bwd_idle), the back substitution of the separator fronts of 8 and 16
    unknowns in one launch (k_sep_bwd_small), and the decision which pairs of a separator level have work (k_fwd_flags + a product dealt from the list).  Same k groups,
    same order of the matrix instructions per accumulator: on a 256^2 grid (49-unknown leaves, levels of >= 256 fronts) with 128 right-hand sides -- point
    sources, an empty block of 64 columns' worth of zeros in the second half, a dense column -- the wavefields are bit for bit those of the library with the
    switch off, and those of the pass that computes every front (arena poisoned with NaNs)."""
    import zephyr_amd as za
    import torch
    n, nrhs = 256, 128
    rng = np.random.default_rng(5)
    c = 1800. + 2000. * rng.random((n, n))
    cfg = dict(nx=n, nz=n, dx=10., dz=10., c=c, freq=7., nPML=8, rtol=1e-10, method='direct', batch=256)
    locs = np.stack([rng.uniform(100., 10. * n - 100., nrhs), rng.uniform(20., 60., nrhs)], axis=1)
    q = za.SparseKaiserSource(cfg)(locs).toarray()
    q[:, 70:128] = 0.0
    q[:, 5] = rng.standard_normal(n * n) + 1j * rng.standard_normal(n * n)
    q = np.ascontiguousarray(q)
    monkeypatch.setenv('HELM_ND_POISON', '1')
    out = {}
    for idle, sparse in (('1', '1'), ('0', '1'), ('1', '0')):
        monkeypatch.setenv('HELM_ND_LEAF_IDLE', idle)
        monkeypatch.setenv('HELM_ND_SPARSE_RHS', sparse)
        op = za.Eurus(cfg)
        R = torch.from_numpy(q).cuda()
        U = torch.empty_like(R)
        op.solveDevice(R.data_ptr(), U.data_ptr(), nrhs, n * n, layout='node')
        torch.cuda.synchronize()
        out[idle + sparse] = U.cpu().numpy()
        assert all(i['status'] == 0 and i['iterations'] == 1 for j, i in enumerate(op.lastInfo) if np.any(q[:, j])), op.lastInfo[:3]
        del op.factors
    assert np.isfinite(out['11']).all()
    assert np.array_equal(out['11'], out['01'])
    assert np.array_equal(out['11'], out['10'])
    assert not np.any(out['11'][:, 100])


@pytest.mark.parametrize('Disc,nf', [('Eurus', 2), ('Eurus', 3), ('MiniZephyr', 4)])
def test_operators_factored_in_the_same_launches_match_one_at_a_time_bit_for_bit(helm_lib, Disc, nf):
    """helm_prefactor_many: the fronts of nf frequencies ride in the same strided batches (batch index = front x frequency, factors interleaved in one buffer).
    Every choice that depends on the batch size is made as for one frequency, so each operator's wavefields must equal those of the one-at-a-time
    factorisation in every bit -- and so must a second solve on the same factors and a solve after one of the set's operators has been destroyed.
    Counterpart: the reference factors one frequency per pool worker (zephyr/backend/distributors.py:161-168, discretization.py:78-103)."""
    import zephyr_amd as za
    from zephyr_amd.models import marmousi_like
    n = 200
    c = marmousi_like(n, n, 12.)
    cls = getattr(za, Disc)
    freqs = [4.0, 6.5, 9.0, 11.0][:nf]
    locs = np.stack([np.linspace(300., 2000., 70), np.full(70, 24.)], axis=1)
    base = dict(nx=n, nz=n, dx=12., dz=12., c=c, nPML=10, rtol=1e-10, method='direct')
    q = za.SparseKaiserSource(base)(locs)
    single = []
    for f in freqs:
        op = cls(dict(base, freq=f))
        op.prefactor()
        single.append(np.array(op * q))
        del op.factors
    ops = [cls(dict(base, freq=f)) for f in freqs]
    za.prefactor_many(ops)
    together = [np.array(op * q) for op in ops]
    for a, b, op in zip(single, together, ops):
        assert np.array_equal(a, b)
        assert all(i['status'] == 0 and i['relres'] <= 1e-10 for i in op.lastInfo)
    del ops[0].factors                                  # the shared buffer lives on for the others
    for a, op in zip(single[1:], ops[1:]):
        assert np.array_equal(a, np.array(op * q))


def test_prefactor_many_falls_back_for_operators_it_cannot_take_together(helm_lib):
    'different grids, a coupled TTI operator, an operator that already has factors: prefactored one by one, results as always'
    import zephyr_amd as za
    from zephyr_amd.models import marmousi_like
    q1 = za.SimpleSource(dict(nx=96, nz=96, dx=10., dz=10.))(np.array([[400., 300.]]))
    q2 = za.SimpleSource(dict(nx=80, nz=112, dx=10., dz=10.))(np.array([[400., 300.]]))
    a = za.Eurus(dict(nx=96, nz=96, dx=10., dz=10., c=marmousi_like(96, 96, 10.), freq=8., rtol=1e-10, method='direct'))
    b = za.Eurus(dict(nx=80, nz=112, dx=10., dz=10., c=marmousi_like(112, 80, 10.), freq=8., rtol=1e-10, method='direct'))
    ua, ub = np.array(a * q1), np.array(b * q2)
    del a.factors, b.factors
    a2 = za.Eurus(dict(nx=96, nz=96, dx=10., dz=10., c=marmousi_like(96, 96, 10.), freq=8., rtol=1e-10, method='direct'))
    b2 = za.Eurus(dict(nx=80, nz=112, dx=10., dz=10., c=marmousi_like(112, 80, 10.), freq=8., rtol=1e-10, method='direct'))
    za.prefactor_many([a2, b2])
    assert np.array_equal(ua, np.array(a2 * q1)) and np.array_equal(ub, np.array(b2 * q2))
    za.prefactor_many([a2, b2])                          # factors are there: a no-op
    assert np.array_equal(ua, np.array(a2 * q1))
