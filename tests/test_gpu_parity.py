"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

from oracle import helm_oracle as ho

pytestmark = pytest.mark.gpu


def hetero_model(nz, nx, seed=3, complex_c=False):
    rng = np.random.default_rng(seed)
    c = 1800. + 2200. * rng.random((nz, nx))
    if complex_c:
        c = c * (1 + 0.01j)
    rho = 1000. + 600. * rng.random((nz, nx))
    return c, rho


def coef_close(got, ref, rtol=1e-13):
    scale = np.abs(ref).max()
    return np.abs(got - ref).max() <= rtol * scale * 50 and np.allclose(got, ref, rtol=1e-11, atol=1e-13 * scale)


@pytest.mark.parametrize('fs', [(False, False, False, False), (True, False, False, True), (False, True, True, False)])
def test_minizephyr_coefficients(helm_lib, fs):
    from zephyr_amd import MiniZephyr
    nz, nx = 40, 50
    c, rho = hetero_model(nz, nx, complex_c=True)
    cfg = dict(nx=nx, nz=nz, dx=7., dz=9., c=c, rho=rho, freq=13., tau=0.4, ky=0.001, nPML=5, freeSurf=fs)
    got = MiniZephyr(cfg).diagonals()[0]
    ref = ho.minizephyr_coefficients(nz, nx, c, rho, 13., dx=7., dz=9., tau=0.4, ky=0.001, nPML=5, freeSurf=fs)
    for k in range(9):
        assert coef_close(got[k], ref[k]), 'plane %d' % k


@pytest.mark.parametrize('aniso', [False, True])
def test_eurus_coefficients(helm_lib, aniso):
    from zephyr_amd import Eurus
    nz, nx = 40, 50
    c, rho = hetero_model(nz, nx, complex_c=True)
    rng = np.random.default_rng(5)
    cfg = dict(nx=nx, nz=nz, dx=7., dz=9., c=c, rho=rho, freq=13., tau=0.4, nPML=6, cPML=800.)
    kw = {}
    if aniso:
        kw = dict(theta=0.3 * rng.random((nz, nx)), eps=0.2 * rng.random((nz, nx)), delta=0.1 * rng.random((nz, nx)))
        cfg.update(kw)
    got = Eurus(cfg).diagonals()
    ref = ho.eurus_coefficients(nz, nx, c, rho, 13., dx=7., dz=9., tau=0.4, nPML=6, cPML=800., **kw)
    for m in range(4):
        for k in range(9):
            assert coef_close(got[m, k], ref[m, k]), 'block %d plane %d' % (m, k)


def test_eurus_blocks_assembled_on_demand(helm_lib):
    """Round 5: an isotropic (eps == delta) Eurus operator is block-triangular and an N-row right-hand side touches M1 alone, so helm_assemble builds M1 only;
    M2 .. M4 come into being when something asks for them.  Order of use must not matter: solve first, then the planes of all four blocks, an apply of
    block 3 and a stacked 2N right-hand side -- each against the oracle (eurus.py:430-464,512-533)."""
    from zephyr_amd import Eurus
    nz, nx = 44, 52
    c, rho = hetero_model(nz, nx)
    rng = np.random.default_rng(8)
    ell = 0.15 * rng.random((nz, nx))                     # elliptical: eps == delta != 0, M3 == 0
    cfg = dict(nx=nx, nz=nz, dx=8., dz=8., c=c, rho=rho, freq=11., nPML=6, cPML=600., eps=ell, delta=ell, rtol=1e-10)
    C4 = ho.eurus_coefficients(nz, nx, c, rho, 11., dx=8., dz=8., nPML=6, cPML=600., eps=ell, delta=ell)
    ref_op = ho.DirectOperator(C4, eurus=True)
    N = nz * nx
    q = np.zeros((N, 2), complex)
    q[(nz // 2) * nx + nx // 3, 0] = 1.; q[7 * nx + 30, 1] = 1j
    op = Eurus(cfg)
    u = op * q                                            # N-row right-hand sides: M1 only
    assert np.linalg.norm(u - ref_op * q) <= 1e-7 * np.linalg.norm(u)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    y3 = op.applyForward(x, block=3)                      # M4 x: the other blocks are built here
    assert np.abs(y3 - ho.stencil_apply(C4[3], x)).max() <= 1e-12 * np.abs(y3).max()
    got = op.diagonals()
    for m in range(4):
        for k in range(9):
            assert coef_close(got[m, k], C4[m, k]), 'block %d plane %d' % (m, k)
    q2 = np.zeros((2 * N, 2), complex)
    q2[:N] = q; q2[N + 9 * nx + 11, 0] = 0.5 - 1j
    op2 = Eurus(cfg)                                      # a fresh operator whose FIRST use is the stacked right-hand side
    u2 = op2 * q2
    r2 = ref_op * q2
    assert u2.shape == r2.shape and np.linalg.norm(u2 - r2) <= 1e-7 * np.linalg.norm(r2)
    u1 = op2 * q                                          # and the N-row solve afterwards, on the same handle
    assert np.linalg.norm(u1 - ref_op * q) <= 1e-7 * np.linalg.norm(u1)


@pytest.mark.parametrize('shape', [(40, 50), (64, 64), (70, 130), (129, 67)])
@pytest.mark.parametrize('nrhs', [1, 5])
def test_apply_matches_oracle(helm_lib, shape, nrhs):
    from zephyr_amd import MiniZephyr
    nz, nx = shape
    c, rho = hetero_model(nz, nx)
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, rho=rho, freq=9.)
    op = MiniZephyr(cfg)
    rng = np.random.default_rng(1234)
    X = rng.standard_normal((nz * nx, nrhs)) + 1j * rng.standard_normal((nz * nx, nrhs))
    C = ho.minizephyr_coefficients(nz, nx, c, rho, 9., dx=10., dz=10.)
    ref = ho.stencil_apply(C, X)
    got = op.applyForward(X)
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    # adjoint: compare with the conjugate-transposed sparse matrix
    A = ho.coefficients_to_csr(C)
    refH = A.conj().T @ X
    gotH = op.applyForward(X, adjoint=True)
    assert np.abs(gotH - refH).max() <= 1e-12 * np.abs(refH).max()


def test_minizephyr_solve_vs_lu(helm_lib):
    from zephyr_amd import MiniZephyr, SimpleSource
    nz = nx = 96
    c, rho = hetero_model(nz, nx)
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, rho=rho, freq=10., rtol=1e-10)
    q = SimpleSource(cfg)(np.array([[300., 320.], [650., 400.], [480., 100.]]))
    op = MiniZephyr(cfg)
    u = op * q
    C = ho.minizephyr_coefficients(nz, nx, c, rho, 10., dx=10., dz=10.)
    ref = ho.DirectOperator(C) * q
    for s in range(q.shape[1]):
        err = np.linalg.norm(u[:, s] - ref[:, s]) / np.linalg.norm(ref[:, s])
        assert err <= 1e-7, (s, err, op.lastInfo)
    assert all(i['relres'] <= 1e-10 for i in op.lastInfo)


def test_eurus_solve_vs_lu(helm_lib):
    from zephyr_amd import Eurus, EurusHD, SimpleSource, StackedSimpleSource
    nz = nx = 96
    c, rho = hetero_model(nz, nx)
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, rho=rho, freq=10., rtol=1e-10)
    locs = np.array([[300., 320.], [650., 400.]])
    q = SimpleSource(cfg)(locs)
    C4 = ho.eurus_coefficients(nz, nx, c, rho, 10., dx=10., dz=10.)
    ref = ho.DirectOperator(C4, eurus=True) * q
    u = Eurus(cfg) * q
    assert u.shape == q.shape
    assert np.linalg.norm(u - ref) / np.linalg.norm(ref) <= 1e-7
    # stacked 2N right-hand side returns both fields
    q2 = StackedSimpleSource(cfg)(locs)
    ref2 = ho.DirectOperator(C4, eurus=True) * q2
    u2 = Eurus(cfg) * q2
    assert u2.shape == q2.shape
    assert np.linalg.norm(u2 - ref2) / np.linalg.norm(ref2) <= 1e-7
    # HD premul
    refhd = ho.DirectOperator(C4, premul=ho.premul_hd(10.), eurus=True) * q
    uhd = EurusHD(cfg) * q
    assert np.linalg.norm(uhd - refhd) / np.linalg.norm(refhd) <= 1e-7
