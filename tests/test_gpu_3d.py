"""GPU: the 3-D 27-point operator (BASELINE config 5; no reference counterpart -> parity against this project's
own CPU statement oracle/helm3d_oracle.py, sparse LU at a small size, and the analytic 3-D Green's function)."""
import numpy as np
import pytest

from oracle import helm3d_oracle as h3

pytestmark = pytest.mark.gpu


def nrm(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def test_3d_coefficients_and_apply_match_oracle(helm_lib):
    import zephyr_amd as za
    nz, ny, nx = 20, 26, 70                  # x not a multiple of the 64-wide tile, y not a multiple of 4
    rng = np.random.default_rng(5)
    c = (1800. + 1500. * rng.random((nz, ny, nx))) * (1 + 0.01j)
    rho = 1000. + 500. * rng.random((nz, ny, nx))
    cfg = dict(nx=nx, ny=ny, nz=nz, dx=9., dy=11., dz=10., c=c, rho=rho, freq=11., tau=0.5, nPML=6, cPML=250.)
    op = za.Helm3D(cfg)
    C = h3.helm3d_coefficients(nz, ny, nx, c, rho, 11., dx=9., dy=11., dz=10., tau=0.5, nPML=6, cPML=250.)
    got = op.diagonals()
    scale = np.abs(C).max()
    assert np.abs(got - C).max() <= 1e-12 * scale
    X = rng.standard_normal((nz * ny * nx, 3)) + 1j * rng.standard_normal((nz * ny * nx, 3))
    ref = h3.stencil_apply3(C, X)
    assert np.abs(op.applyForward(X) - ref).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize('nrhs', [4, 9, 16])
def test_3d_apply_with_coefficients_rebuilt_on_the_fly(helm_lib, monkeypatch, nrhs):
    """Round 5 (SURVEY.md 7 hard-part 6): from 4 right-hand sides up the 27-point apply rebuilds its coefficients from K = om^2 / (rho c^2),
    b = 1 / rho (24 B per point through the tile's LDS) and the three per-axis factor tables instead of reading 27 stored planes (432 B per point).  Against
    oracle/helm3d_oracle.py to 1e-12 on a heterogeneous, complex-velocity model with anisotropic spacing and damping, and BIT FOR BIT the stored-plane apply
    (HELM_MG3_OTF=0): both build a coefficient with the same contraction-pinned function."""
    import zephyr_amd as za
    nz, ny, nx = 20, 26, 70                  # x not a multiple of the 64-wide tile, y not a multiple of 4
    rng = np.random.default_rng(50 + nrhs)
    c = (1800. + 1500. * rng.random((nz, ny, nx))) * (1 + 0.01j)
    rho = 1000. + 500. * rng.random((nz, ny, nx))
    cfg = dict(nx=nx, ny=ny, nz=nz, dx=9., dy=11., dz=10., c=c, rho=rho, freq=11., tau=0.5, nPML=6, cPML=250.)
    C = h3.helm3d_coefficients(nz, ny, nx, c, rho, 11., dx=9., dy=11., dz=10., tau=0.5, nPML=6, cPML=250.)
    X = rng.standard_normal((nz * ny * nx, nrhs)) + 1j * rng.standard_normal((nz * ny * nx, nrhs))
    ref = h3.stencil_apply3(C, X)
    out = {}
    for mode in ('2', '0'):                    # 2: on the fly whatever the batch width (1, the default: from 4 right-hand sides per workgroup up); 0: stored planes
        monkeypatch.setenv('HELM_MG3_OTF', mode)
        op = za.Helm3D(cfg)
        assert np.abs(op.diagonals() - C).max() <= 1e-12 * np.abs(C).max()
        out[mode] = op.applyForward(X)
        assert np.abs(out[mode] - ref).max() <= 1e-12 * np.abs(ref).max(), mode
    assert np.array_equal(out['2'], out['0'])                   # same kernel body, coefficients rebuilt bit for bit


def test_3d_solve_matches_sparse_lu(helm_lib):
    import zephyr_amd as za
    n = 22
    rng = np.random.default_rng(6)
    c = 2000. + 600. * rng.random((n, n, n))
    cfg = dict(nx=n, ny=n, nz=n, dx=10., c=c, rho=1., freq=14., nPML=5, cPML=200., rtol=1e-10, maxit=50000)
    q = np.zeros((n ** 3, 2), complex)
    q[(11 * n + 11) * n + 11, 0] = 1.
    q[(8 * n + 13) * n + 9, 1] = 1j
    op = za.Helm3D(cfg)
    u = op * q
    C = h3.helm3d_coefficients(n, n, n, c, 1., 14., dx=10., nPML=5, cPML=200.)
    ref = h3.DirectOperator3(C) * q
    assert nrm(u, ref) <= 1e-7, op.lastInfo
    assert all(i['relres'] <= 1e-10 for i in op.lastInfo)


def test_3d_analytic_green_function(helm_lib):
    """homogeneous c = 2000 m/s, interior window: |u - G| / |G| < 5e-2 with G = -rho h^3 e^{ikr} / (4 pi r)"""
    import zephyr_amd as za
    n, dx, f, c0 = 64, 10., 12., 2000.
    cfg = dict(nx=n, ny=n, nz=n, dx=dx, c=c0, rho=1., freq=f, nPML=10, rtol=1e-8, maxit=100000)
    q = np.zeros(n ** 3, complex)
    s = n // 2
    q[(s * n + s) * n + s] = 1.
    u = (za.Helm3D(cfg) * q).reshape((n, n, n))
    iz, iy, ix = np.mgrid[0:n, 0:n, 0:n]
    r = dx * np.sqrt((iz - s) ** 2 + (iy - s) ** 2 + (ix - s) ** 2)
    m = (r > 3 * dx) & (iz > 12) & (iz < n - 13) & (iy > 12) & (iy < n - 13) & (ix > 12) & (ix < n - 13)
    g = h3.green3d(2 * np.pi * f / c0, r[m], 1.0, dx ** 3)
    assert np.linalg.norm(u[m] - g) / np.linalg.norm(g) < 5e-2


def test_3d_multigrid_preconditioner(helm_lib):
    """method='mg' on the 3-D operator (mg3d.hip): same wavefield as the sparse LU, far fewer iterations than Jacobi-BiCGSTAB"""
    import zephyr_amd as za
    from oracle import helm3d_oracle as h3
    import scipy.sparse.linalg as spla
    nz, ny, nx = 30, 26, 28          # (the sparse LU of a 3-D grid is the slow part of this test)
    rng = np.random.default_rng(3)
    c = 1900. + 400. * rng.random((nz, ny, nx))
    rho = 1000. + 100. * rng.random((nz, ny, nx))
    cfg = dict(nx=nx, ny=ny, nz=nz, dx=10., c=c, rho=rho, freq=6., nPML=6, rtol=1e-9, maxit=20000)
    N = nz * ny * nx
    q = np.zeros((N, 2), complex)
    q[(15 * ny + 12) * nx + 14, 0] = 1.0
    q[(9 * ny + 18) * nx + 20, 1] = 1.0 - 0.5j
    A = h3.coefficients_to_csr3(h3.helm3d_coefficients(nz, ny, nx, c, rho, 6., dx=10., nPML=6)).tocsc()
    ref = np.conj(spla.splu(A).solve(q))
    op_j = za.Helm3D(dict(cfg, method='bicgstab'))
    uj = op_j * q
    op_m = za.Helm3D(dict(cfg, method='mg'))
    um = op_m * q
    assert np.linalg.norm(um - ref) / np.linalg.norm(ref) <= 1e-6, op_m.lastInfo
    its_j = max(i['iterations'] for i in op_j.lastInfo); its_m = max(i['iterations'] for i in op_m.lastInfo)
    assert all(i['status'] == 0 and i['method'] == 3 for i in op_m.lastInfo)
    assert its_m * 2 < its_j, (its_m, its_j)
    print('3-D iterations: multigrid %d, Jacobi %d' % (its_m, its_j))


def test_3d_layer_preserving_hierarchy(helm_lib, monkeypatch):
    """Oversampled grid (45 points per wavelength: two layer-preserving coarsenings, block-tridiagonal direct solve of the third level):
    same wavefield as the sparse LU on a layered model with a density contrast, an order of magnitude fewer iterations than the
    standard shifted cycle (HELM_MG3_KEEP=0) it replaces there."""
    import zephyr_amd as za
    from oracle import helm3d_oracle as h3
    import scipy.sparse.linalg as spla
    nz, ny, nx, f = 30, 32, 28, 4.
    iz = np.arange(nz)[:, None, None]
    c = (1800. + 25. * iz + 150. * (iz > 18)) * np.ones((nz, ny, nx))
    rho = 1000. + 300. * (iz > 18) * np.ones((nz, ny, nx))
    cfg = dict(nx=nx, ny=ny, nz=nz, dx=10., c=c, rho=rho, freq=f, nPML=6, rtol=1e-9, maxit=20000, method='mg')
    N = nz * ny * nx
    q = np.zeros((N, 3), complex)
    q[(15 * ny + 12) * nx + 14, 0] = 1.0
    q[(9 * ny + 18) * nx + 20, 1] = 1.0 - 0.5j
    q[(22 * ny + 27) * nx + 4, 2] = 2.0j                 # inside the absorbing layers
    A = h3.coefficients_to_csr3(h3.helm3d_coefficients(nz, ny, nx, c, rho, f, dx=10., nPML=6)).tocsc()
    ref = np.conj(spla.splu(A).solve(q))
    op = za.Helm3D(cfg)
    u = op * q
    assert all(i['status'] == 0 and i['method'] == 3 for i in op.lastInfo), op.lastInfo
    assert np.linalg.norm(u - ref) / np.linalg.norm(ref) <= 1e-6
    its_keep = max(i['iterations'] for i in op.lastInfo)
    monkeypatch.setenv('HELM_MG3_KEEP', '0')
    op0 = za.Helm3D(cfg)
    u0 = op0 * q
    its_std = max(i['iterations'] for i in op0.lastInfo)
    assert np.linalg.norm(u0 - ref) / np.linalg.norm(ref) <= 1e-6
    print('3-D iterations: layer-preserving hierarchy %d, standard shifted cycle %d' % (its_keep, its_std))
    assert its_keep <= 30 and its_keep * 4 < its_std, (its_keep, its_std)


def test_3d_layer_preserving_hierarchy_anisotropic_spacing_and_attenuation(helm_lib):
    """The same hierarchy with dx != dy != dz and a finite tau (complex omega in the mass term and in the layer stretch): parity with the
    sparse LU of the oracle's matrix; more than 16 right-hand sides in one batch go through the generic batched GEMM of the coarse solve."""
    import zephyr_amd as za
    from oracle import helm3d_oracle as h3
    import scipy.sparse.linalg as spla
    nz, ny, nx, f, tau = 26, 28, 30, 4., 0.8
    dx, dy, dz = 10., 9., 11.
    rng = np.random.default_rng(11)
    c = 1900. + 300. * rng.random((nz, ny, nx))
    cfg = dict(nx=nx, ny=ny, nz=nz, dx=dx, dy=dy, dz=dz, c=c, rho=1000., freq=f, tau=tau, nPML=6, rtol=1e-9, maxit=5000, method='mg')
    N = nz * ny * nx
    q = np.zeros((N, 2), complex)
    q[(13 * ny + 12) * nx + 14, 0] = 1.0
    q[(9 * ny + 17) * nx + 20, 1] = 1.0 - 0.5j
    A = h3.coefficients_to_csr3(h3.helm3d_coefficients(nz, ny, nx, c, 1000., f, dx=dx, dy=dy, dz=dz, tau=tau, nPML=6)).tocsc()
    lu = spla.splu(A)
    ref = np.conj(lu.solve(q))
    op = za.Helm3D(cfg)
    u = op * q
    assert all(i['status'] == 0 and i['method'] == 3 for i in op.lastInfo), op.lastInfo
    assert np.linalg.norm(u - ref) / np.linalg.norm(ref) <= 1e-6
    assert max(i['iterations'] for i in op.lastInfo) <= 30, op.lastInfo
    # a batch of 20: the plane products of the coarse solve leave k_bt_apply (16 right-hand sides) for the batched GEMM
    q20 = np.zeros((N, 20), complex)
    for s in range(20):
        q20[((4 + s) * ny + 5 + s) * nx + 6 + s, s] = 1.0 + 0.1j * s
    op20 = za.Helm3D(dict(cfg, batch=20))
    u20 = op20 * q20
    assert all(i['status'] == 0 for i in op20.lastInfo), op20.lastInfo
    ref20 = np.conj(lu.solve(q20))
    assert np.linalg.norm(u20 - ref20) / np.linalg.norm(ref20) <= 1e-6
    assert max(i['iterations'] for i in op20.lastInfo) <= 30, op20.lastInfo


def test_3d_retreat_to_the_standard_cycle(helm_lib, monkeypatch):
    """If the layer-preserving hierarchy has not converged within its iteration cap (forced here: 3), the frequency retreats to the
    standard cycle and goes on from the iterates reached: same wavefield, status 0."""
    import zephyr_amd as za
    nz, ny, nx = 30, 26, 28
    rng = np.random.default_rng(3)
    c = 1900. + 400. * rng.random((nz, ny, nx))
    cfg = dict(nx=nx, ny=ny, nz=nz, dx=10., c=c, rho=1000., freq=6., nPML=6, rtol=1e-9, maxit=20000, method='auto')
    N = nz * ny * nx
    q = np.zeros((N, 2), complex)
    q[(15 * ny + 12) * nx + 14, 0] = 1.0
    q[(9 * ny + 18) * nx + 20, 1] = 1.0 - 0.5j
    op = za.Helm3D(cfg)
    u = op * q
    its = max(i['iterations'] for i in op.lastInfo)
    monkeypatch.setenv('HELM_MG3_KEEP_CAP', '3')
    op2 = za.Helm3D(cfg)
    u2 = op2 * q
    assert all(i['status'] == 0 and i['relres'] <= 1e-9 for i in op2.lastInfo), op2.lastInfo
    its2 = max(i['iterations'] for i in op2.lastInfo)
    assert its2 > 3 * its, (its, its2)                     # it did go on with the slower cycle
    assert np.linalg.norm(u2 - u) / np.linalg.norm(u) <= 1e-7
    u3 = op2 * q                                           # the retreat holds for the rest of this frequency
    assert all(i['status'] == 0 for i in op2.lastInfo)
    assert np.linalg.norm(u3 - u) / np.linalg.norm(u) <= 1e-7


def test_3d_multifreq_dispatcher(helm_lib):
    """The multi-frequency dispatcher (zephyr/backend/dispatcher.py:160-187 contract) over the 3-D operator: one sub-problem per frequency,
    every wavefield satisfies its own operator."""
    import zephyr_amd as za
    nz, ny, nx = 30, 32, 28
    c = 2000. * np.ones((nz, ny, nx))
    cfg = dict(nx=nx, ny=ny, nz=nz, dx=10., c=c, rho=1000., freqs=[3., 4.], nPML=6, rtol=1e-8, Disc=za.Helm3D, method='auto')
    mf = za.MultiFreq(cfg)
    N = nz * ny * nx
    q = np.zeros((N, 2), complex)
    q[(15 * ny + 12) * nx + 14, 0] = 1.
    q[(9 * ny + 18) * nx + 20, 1] = 1j
    us = list(mf * q)
    assert len(us) == 2 and all(u.shape == (N, 2) for u in us)
    for f, u, sub in zip(cfg['freqs'], us, mf.subProblems):
        assert all(i['status'] == 0 for i in sub.lastInfo), sub.lastInfo
        op = za.Helm3D(dict(cfg, freq=f))
        r = op.applyForward(u.conj()) - q
        assert np.linalg.norm(r) <= 2e-8 * np.linalg.norm(q)


def test_3d_mid_size_properties_128x128x64(helm_lib):
    """Config-5 geometry at one eighth of its size (c = 2000 m/s, h = 10 m, 5 Hz): size-independent properties of the solve --
    residual of the returned field through the independent apply entry point, conj-linearity, agreement with the analytic
    3-D Green's function on an interior window (zephyr/backend/analytical.py:55-59 is the reference's own analytic pin)."""
    import zephyr_amd as za
    nz, ny, nx, dx, f, c0 = 64, 128, 128, 10., 5., 2000.
    cfg = dict(nx=nx, ny=ny, nz=nz, dx=dx, c=c0, rho=1., freq=f, nPML=10, rtol=1e-8, maxit=60000, method='auto', batch=2)
    N = nz * ny * nx
    sz, sy, sx = 30, 64, 60
    q = np.zeros((N, 2), complex)
    q[(sz * ny + sy) * nx + sx, 0] = 1.
    q[((sz + 6) * ny + sy - 20) * nx + sx + 25, 1] = 1j
    op = za.Helm3D(cfg)
    u = op * q
    assert all(i['status'] == 0 and i['relres'] <= 1e-8 for i in op.lastInfo), op.lastInfo
    r = op.applyForward(u.conj()) - q                              # A conj(u) = q
    assert np.linalg.norm(r, axis=0).max() <= 2e-8 * np.linalg.norm(q, axis=0).min()
    usum = op * (q[:, 0] - 2j * q[:, 1])
    assert nrm(usum, u[:, 0] + 2j * u[:, 1]) <= 1e-6               # conj-linear
    iz, iy, ix = np.mgrid[0:nz, 0:ny, 0:nx]
    rr = dx * np.sqrt((iz - sz) ** 2 + (iy - sy) ** 2 + (ix - sx) ** 2)
    m = (rr > 3 * dx) & (iz > 12) & (iz < nz - 13) & (iy > 12) & (iy < ny - 13) & (ix > 12) & (ix < nx - 13)
    g = h3.green3d(2 * np.pi * f / c0, rr[m], 1.0, dx ** 3)
    assert np.linalg.norm(u[:, 0].reshape((nz, ny, nx))[m] - g) / np.linalg.norm(g) < 5e-2
    print('3-D 128x128x64 @ 5 Hz: iterations %s' % [i['iterations'] for i in op.lastInfo])
