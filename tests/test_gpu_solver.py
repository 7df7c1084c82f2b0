"""GPU: solver behaviour through the C ABI -- every Krylov method against sparse LU, the reference's
own test configurations and golden wavefields, dispatcher / survey / gradient against goldens, edge
cases and error conventions, and size-independent properties at the BASELINE grid size."""
import os
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import helm_oracle as ho

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def load(name):
    return np.load(os.path.join(GOLD, name))


def nrm(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def hetero(nz, nx, seed=3):
    rng = np.random.default_rng(seed)
    return 1800. + 2200. * rng.random((nz, nx)), 1000. + 600. * rng.random((nz, nx))


@pytest.mark.parametrize('method', ['bicgstab', 'cgnr', 'mg', 'auto'])
@pytest.mark.parametrize('cls', ['MiniZephyr', 'Eurus'])
def test_every_method_matches_sparse_lu(helm_lib, method, cls):
    import zephyr_amd as za
    nz, nx = 70, 90                          # not multiples of the 64 x 8 tile
    c, rho = hetero(nz, nx)
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, rho=rho, freq=9., nPML=8, rtol=1e-10, method=method, maxit=400000)
    q = za.SimpleSource(cfg)(np.array([[300., 320.], [650., 400.]]))
    op = getattr(za, cls)(cfg)
    u = op * q
    if cls == 'MiniZephyr':
        ref = ho.DirectOperator(ho.minizephyr_coefficients(nz, nx, c, rho, 9., dx=10., dz=10., nPML=8)) * q
    else:
        ref = ho.DirectOperator(ho.eurus_coefficients(nz, nx, c, rho, 9., dx=10., dz=10., nPML=8), eurus=True) * q
    assert nrm(u, ref) <= 1e-7, op.lastInfo
    assert all(i['status'] == 0 and i['relres'] <= 1e-10 for i in op.lastInfo)


def test_reference_test_configurations_and_analytic_thresholds(helm_lib):
    """zephyr/backend/Tests/test_MiniZephyr.py:81-114 and test_Eurus.py:39-151 on the GPU path."""
    import zephyr_amd as za
    g = load('g3_wavefields.npz')
    nx, nz = 100, 200
    iz, ix = g['rec_iz'], int(g['rec_ix'])
    sloc = np.array([[25., 25.]])

    def window_error(u, scA):
        uA = za.AnalyticalHelmholtz(scA)(sloc).reshape((nz, nx))
        seg = (uA[40:180, 40:80] - u[40:180, 40:80]) / abs(uA[40:180, 40:80])
        return abs(np.sqrt((seg.conj() * seg).sum()) / seg.size)

    sc = dict(c=2500., rho=1., nx=nx, nz=nz, freq=2e2)
    u = (za.MiniZephyr(sc) * za.SimpleSource(sc)(sloc))[:, 0].reshape((nz, nx))
    assert nrm(u[iz, ix], g['mz_line']) <= 1e-7
    assert window_error(u, sc) < 1e-2
    uhd = (za.MiniZephyrHD(sc) * za.SimpleSource(sc)(sloc))[:, 0].reshape((nz, nx))
    assert nrm(uhd[iz, ix], g['mzhd_line']) <= 1e-7

    sce = dict(c=2000. * np.ones((nz, nx)), rho=np.ones((nz, nx)), nx=nx, nz=nz, dx=1, dz=1, freq=2e2, nPML=10, cPML=1e3,
               freeSurf=[False, False, False, False])
    q2 = za.StackedSimpleSource(sce)(sloc)
    ue = za.Eurus(sce) * q2
    assert ue.shape == (2 * nx * nz, 1)
    ue = ue[:nx * nz, 0].reshape((nz, nx))
    assert nrm(ue[iz, ix], g['eu_line']) <= 1e-7
    assert window_error(ue, dict(sce, c=2000., rho=1.)) < 3e-2
    uehd = (za.EurusHD(sce) * q2)[:nx * nz, 0].reshape((nz, nx))
    assert nrm(uehd[iz, ix], g['euhd_line']) <= 1e-7
    scl = dict(sce, theta=np.zeros((nz, nx)), eps=0.2 * np.ones((nz, nx)), delta=0.2 * np.ones((nz, nx)))
    uel = (za.Eurus(scl) * q2)[:nx * nz, 0].reshape((nz, nx))
    assert nrm(uel[iz, ix], g['euell_line']) <= 1e-7
    assert window_error(uel, dict(scl, c=2000., rho=1., eps=0.2, theta=0.)) < 3e-2


def test_golden_heterogeneous_fields_and_cfg1(helm_lib):
    import zephyr_amd as za
    g = load('g3_wavefields.npz')
    sc = dict(nx=64, nz=64, dx=10., dz=10., c=g['het_c'], rho=g['het_rho'], freq=12., nPML=8)
    assert nrm(za.MiniZephyr(sc) * g['het_q'], g['het_mz']) <= 1e-7
    assert nrm(za.Eurus(sc) * g['het_q'], g['het_eu']) <= 1e-7
    assert nrm(za.Eurus(sc) * np.vstack([g['het_q'], 0 * g['het_q']]), g['het_eu_stacked']) <= 1e-7
    sc1 = dict(nx=128, nz=128, dx=10., dz=10., c=2000., rho=1., nPML=10, freq=5.)          # BASELINE config 1
    u1 = za.MiniZephyr(sc1) * za.SimpleSource(sc1)(np.array([[640., 640.]]))
    assert nrm(u1[:, 0], g['cfg1_u']) <= 1e-7


def test_eurus_stacked_rhs_with_second_field_source(helm_lib):
    """2N right-hand side with a non-zero second half exercises v = M4^-1 q2, u = M1^-1 (q1 - M2 v)."""
    import zephyr_amd as za
    nz = nx = 48
    c, rho = hetero(nz, nx, 9)
    rng = np.random.default_rng(2)
    delta = 0.1 * rng.random((nz, nx))
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, rho=rho, freq=11., nPML=6, eps=delta, delta=delta, theta=0.2 * rng.random((nz, nx)))
    q = np.zeros((2 * nz * nx, 2), complex)
    q[20 * nx + 22, 0] = 1.; q[nz * nx + 25 * nx + 30, 0] = 0.5j; q[nz * nx + 10 * nx + 12, 1] = 1.
    u = za.Eurus(cfg) * q
    C4 = ho.eurus_coefficients(nz, nx, c, rho, 11., dx=10., dz=10., nPML=6, eps=delta, delta=delta, theta=cfg['theta'])
    ref = ho.DirectOperator(C4, eurus=True) * q
    assert nrm(u, ref) <= 1e-7


def test_multifreq_on_gpu_matches_golden(helm_lib):
    import zephyr_amd as za
    g = load('g4_multifreq.npz')
    nz, nx = g['c'].shape
    sc = dict(nx=nx, nz=nz, dx=10., dz=10., c=g['c'], rho=g['rho'], nPML=6, freqs=list(g['freqs']), Disc=za.MiniZephyr,
              scaleTerm=0.5 - 0.25j)
    mf = za.MultiFreq(sc)
    assert nrm(np.stack(list(mf * g['q'])), g['shared']) <= 1e-7
    qlist = [g['q'] * (1 + i) for i in range(3)]
    assert nrm(np.stack(list(mf * qlist)), g['list']) <= 1e-7
    assert mf.factors is True
    del mf.factors
    assert mf.factors is False
    vm = za.ViscoMultiFreq(dict(sc, Q=g['Q'], freqBase=10., scaleTerm=1.))
    assert nrm(np.stack(list(vm * g['q'])), g['visco']) <= 1e-7


def test_dpred_and_gradient_on_gpu_match_golden(helm_lib):
    import zephyr_amd as za
    from zephyr_amd.problem import Helm2DProblem
    from zephyr_amd.survey import Helm2DSurvey
    g = load('g6_survey.npz')
    nz, nx = g['c'].shape
    sc = dict(nx=nx, nz=nz, dx=10., dz=10., c=g['c'], rho=g['rho'], nPML=6, freqs=list(g['freqs']), Disc=za.MiniZephyrHD,
              sterms=g['sterms'], geom=dict(src=g['src'], rec=g['rec'], mode='fixed'))
    prob, surv = Helm2DProblem(sc), Helm2DSurvey(sc)
    prob.pair(surv)
    assert nrm(surv.dpred(), g['dpred']) <= 1e-7
    assert nrm(prob.Jtvec(None, g['resid']), g['g_mux']) <= 1e-6
    uF = prob.fields()
    assert nrm(prob.Jtvec(None, g['resid'], u=uF), g['g_u']) <= 1e-6


def test_edge_cases_and_error_conventions(helm_lib):
    import zephyr_amd as za
    from zephyr_amd._lib import HelmError
    nz = nx = 40
    cfg = dict(nx=nx, nz=nz, dx=10., c=2500., freq=10.)
    op = za.MiniZephyr(cfg)
    assert np.all(op * np.zeros((nz * nx, 2), complex) == 0)                 # zero right-hand side
    assert (op * np.eye(nz * nx, 1, -820)[:, 0].astype(complex)).shape == (nz * nx,)        # 1-D in, 1-D out
    sparse_q = sp.csc_matrix(([1.], ([820], [0])), shape=(nz * nx, 1))
    assert nrm(op * sparse_q, (op * sparse_q.toarray())) == 0               # sparse RHS densified, bitwise reproducible
    with pytest.raises(ValueError):
        op * np.zeros((nz * nx + 1, 1), complex)
    with pytest.raises(ValueError):
        za.Eurus(cfg) * np.zeros((3 * nz * nx, 1), complex)
    rng = np.random.default_rng(0)
    tti = dict(cfg, eps=0.2 * rng.random((nz, nx)), delta=0.1 * rng.random((nz, nx)), method='mg')
    with pytest.raises(HelmError) as ei:                                       # the coupled system has no multigrid path
        za.Eurus(tti) * np.ones((nz * nx, 1), complex)
    assert 'UNSUPPORTED' in str(ei.value)
    assert za.Eurus(tti).diagonals().shape == (4, 9, nz, nx)
    capped = za.MiniZephyr(dict(cfg, maxit=20, method='bicgstab'))
    with pytest.raises(ArithmeticError):
        capped * np.eye(nz * nx, 1, -820)[:, 0].astype(complex)
    assert capped.lastInfo[0]['status'] == 1 and capped.lastInfo[0]['iterations'] >= 20


def test_full_size_properties_1024(helm_lib):
    """BASELINE grid size: residual of the returned field through the independent apply entry point,
    linearity, and run-to-run bit reproducibility."""
    import zephyr_amd as za
    from zephyr_amd.models import marmousi_like
    n, dx = 1024, 9.
    c = marmousi_like(n, n, dx)
    cfg = dict(nx=n, nz=n, dx=dx, dz=dx, c=c, freq=5., rtol=1e-10)
    op = za.Eurus(cfg)
    locs = np.array([[3000., 20.], [6100., 20.]])
    q = za.SparseKaiserSource(cfg)(locs).toarray()
    u = op * q
    r = op.applyForward(u.conj()) - q                       # A conj(u) = q   (discretization.py:101-103)
    assert np.linalg.norm(r, axis=0).max() / np.linalg.norm(q, axis=0).min() <= 2e-10
    usum = op * (q[:, :1] + 2j * q[:, 1:])
    assert nrm(usum, u[:, :1] - 2j * u[:, 1:]) <= 1e-7      # conj-linear: conj(A^-1 (q1 + 2i q2))
    u2 = op * q
    assert np.array_equal(u, u2)
    assert max(i['iterations'] for i in op.lastInfo) < 2000


def test_device_resident_gradient_equals_host_gradient(helm_lib):
    """Jtvec mux branch through helm_imaging_accumulate_device (wavefields never leave HBM) == numpy imaging."""
    import zephyr_amd as za
    from zephyr_amd.problem import Helm2DProblem
    from zephyr_amd.survey import Helm2DSurvey
    g = load('g6_survey.npz')
    nz, nx = g['c'].shape
    base = dict(nx=nx, nz=nz, dx=10., dz=10., c=g['c'], rho=g['rho'], nPML=6, freqs=list(g['freqs']), Disc=za.MiniZephyrHD,
                sterms=g['sterms'], geom=dict(src=g['src'], rec=g['rec'], mode='fixed'), scaleTerm=0.7 + 0.2j)
    out = []
    for host in (False, True):
        prob, surv = Helm2DProblem(dict(base, hostGradient=host)), Helm2DSurvey(base)
        prob.pair(surv)
        assert prob._deviceGradientAvailable() is (not host)
        out.append(prob.Jtvec(None, g['resid']))
    assert nrm(out[0], out[1]) <= 1e-9
    assert nrm(out[0] / (0.7 + 0.2j) ** 2, g['g_mux']) <= 1e-6


def test_minizephyr25d_reference_configuration(helm_lib):
    """test_MiniZephyr.py:116-152 (nky=20 vs the analytic 3-D Green's function) and :35-56 (nky=4)."""
    import zephyr_amd as za
    g = load('g8_25d.npz')
    nx, nz = 100, 200
    sc = dict(c=2500., rho=1., nx=nx, nz=nz, freq=2e2, nky=20, parallel=False)
    sloc = np.array([[25., 25.]])
    u = (za.MiniZephyr25D(sc) * za.SimpleSource(sc)(sloc))[:, 0].reshape((nz, nx))
    assert nrm(u[np.arange(5, 196, 10), 60], g['line']) <= 1e-7
    uA = za.AnalyticalHelmholtz(dict(sc, **{'3D': True}))(sloc).reshape((nz, nx))
    seg = (uA[40:180, 40:80] - u[40:180, 40:80]) / abs(uA[40:180, 40:80])
    assert abs(np.sqrt((seg.conj() * seg).sum()) / seg.size) < 1e-2
    sc4 = dict(sc, nky=4)
    u4 = (za.MiniZephyr25D(sc4) * za.SimpleSource(sc4)(np.array([[50., 100.]])))[:, 0].reshape((nz, nx))
    assert nrm(u4[np.arange(5, 196, 10), 60], g['nky4_line']) <= 1e-7


def test_eurus_tti_coupled_two_field_system(helm_lib):
    """eps != delta: M3 != 0, the full 2N x 2N system [[M1,M2],[M3,M4]] is solved (eurus.py:430-464); N-row and
    stacked 2N-row right-hand sides against the sparse LU of the reference-identical matrix."""
    import zephyr_amd as za
    nz, nx = 44, 52
    c, rho = hetero(nz, nx, 13)
    rng = np.random.default_rng(14)
    theta = 0.5 * rng.random((nz, nx)) - 0.25
    eps = 0.25 * rng.random((nz, nx))
    delta = 0.12 * rng.random((nz, nx))
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, rho=rho, freq=9., nPML=6, theta=theta, eps=eps, delta=delta, rtol=1e-10, maxit=400000)
    C4 = ho.eurus_coefficients(nz, nx, c, rho, 9., dx=10., dz=10., nPML=6, theta=theta, eps=eps, delta=delta)
    assert np.abs(C4[2]).max() > 0                     # genuinely coupled
    lu = ho.DirectOperator(C4, eurus=True)
    q = za.SimpleSource(cfg)(np.array([[250., 220.], [330., 150.]]))
    op = za.Eurus(cfg)
    u = op * q
    assert u.shape == q.shape
    assert nrm(u, lu * q) <= 1e-7, op.lastInfo
    q2 = np.vstack([q, 0.3j * q[::-1]])
    u2 = op * q2
    assert u2.shape == q2.shape
    assert nrm(u2, lu * q2) <= 1e-7, op.lastInfo
    assert all(i['relres'] <= 1e-10 for i in op.lastInfo)


def test_xhlayr_fixture_on_gpu(helm_lib):
    """the reference's heterogeneous SEG-Y fixture (decoded by oracle/make_golden.py): receiver data at 100 Hz"""
    import zephyr_amd as za
    g = load('g9_xhlayr.npz')
    c = g['c']; nz, nx = c.shape
    sc = dict(nx=nx, nz=nz, dx=1., dz=1., c=c, freq=float(g['freq']))
    q = za.SparseKaiserSource(sc)(g['src'])
    R = za.SparseKaiserSource(sc)(g['rec']).T
    u = za.MiniZephyrHD(sc) * q
    assert nrm(R @ u, g['data']) <= 1e-7
    assert nrm(u[:, 0].reshape((nz, nx))[:, 60], g['u_src0_col60']) <= 1e-7


@pytest.mark.parametrize('case', ['mz_freesurf', 'mz_visco_tau', 'eurus_rect', 'eurus_elliptical', 'mz_dxdz', 'eurus_small_pml'])
def test_default_solver_is_robust_across_configurations(helm_lib, case):
    """method='auto' (multigrid-preconditioned, with its safety net) on configurations the reference supports:
    free surface, complex velocity + Laplace damping, rectangular grids, elliptical anisotropy, dx != dz, thin PML."""
    import zephyr_amd as za
    from zephyr_amd.models import marmousi_like
    rng = np.random.default_rng(77)
    if case == 'mz_freesurf':
        nz, nx = 150, 230
        c = marmousi_like(nz, nx, 10.)
        cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, freq=7., freeSurf=(True, False, False, False))
        cls, oracle = za.MiniZephyr, lambda: ho.minizephyr_coefficients(nz, nx, c, ho.gardner_rho(c), 7., dx=10., dz=10., freeSurf=(True, False, False, False))
    elif case == 'mz_visco_tau':
        nz, nx = 160, 160
        cr = marmousi_like(nz, nx, 8.)
        c = cr + 0.5j * cr / 60.
        cfg = dict(nx=nx, nz=nz, dx=8., dz=8., c=c, rho=2000., freq=9., tau=1.5)
        cls, oracle = za.MiniZephyrHD, lambda: ho.minizephyr_coefficients(nz, nx, c, 2000., 9., dx=8., dz=8., tau=1.5)
    elif case == 'eurus_rect':
        nz, nx = 120, 330
        c = marmousi_like(nz, nx, 12.)
        cfg = dict(nx=nx, nz=nz, dx=12., dz=12., c=c, freq=6.)
        cls, oracle = za.Eurus, lambda: ho.eurus_coefficients(nz, nx, c, ho.gardner_rho(c), 6., dx=12., dz=12.)
    elif case == 'eurus_elliptical':
        nz, nx = 140, 140
        c = marmousi_like(nz, nx, 10.)
        d = 0.15 * rng.random((nz, nx)); th = 0.4 * rng.random((nz, nx)) - 0.2
        cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, freq=8., eps=d, delta=d, theta=th)
        cls, oracle = za.Eurus, lambda: ho.eurus_coefficients(nz, nx, c, ho.gardner_rho(c), 8., dx=10., dz=10., eps=d, delta=d, theta=th)
    elif case == 'mz_dxdz':
        nz, nx = 130, 170
        c = marmousi_like(nz, nx, 10.)
        cfg = dict(nx=nx, nz=nz, dx=12., dz=8., c=c, rho=1800., freq=8., nPML=14)
        cls, oracle = za.MiniZephyr, lambda: ho.minizephyr_coefficients(nz, nx, c, 1800., 8., dx=12., dz=8., nPML=14)
    else:
        nz, nx = 128, 128
        c = marmousi_like(nz, nx, 10.)
        cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, freq=10., nPML=5, cPML=400.)
        cls, oracle = za.Eurus, lambda: ho.eurus_coefficients(nz, nx, c, ho.gardner_rho(c), 10., dx=10., dz=10., nPML=5, cPML=400.)
    locs = np.array([[0.3 * nx * cfg['dx'], 0.15 * nz * cfg['dz']], [0.7 * nx * cfg['dx'], 0.6 * nz * cfg['dz']]])
    q = za.SparseKaiserSource(cfg)(locs).toarray()
    op = cls(cfg)
    u = op * q
    C = oracle()
    premul = complex(op.premul)
    ref = ho.DirectOperator(C, premul=premul, eurus=(cls is za.Eurus)) * q
    assert nrm(u, ref) <= 1e-7, op.lastInfo
    assert max(i['iterations'] for i in op.lastInfo) < 20000, op.lastInfo
