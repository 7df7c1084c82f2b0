#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (dev container only)  --  test infrastructure.

Imports /root/reference/zephyr (read-only) through the stand-in packages in oracle/refshim and
(1) checks oracle/helm_oracle.py entry-by-entry against it, (2) writes small golden input/output
vectors that travel with the repo.  Only data is written: inputs and the reference's outputs.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

Nothing here runs on the GPU box and the product package never imports it.
"""
import os
import sys
import warnings

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
sys.path[:0] = [os.path.join(HERE, 'refshim'), REF, ROOT]
warnings.simplefilter('ignore')

import numpy as np                                      # noqa: E402
import scipy.sparse as sp                               # noqa: E402
import zephyr.backend as zb                             # noqa: E402  (the reference)
from oracle import helm_oracle as ho                    # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)


def planes_from_matrix(A, nz, nx):
    """Reference sparse matrix (N x N) -> C[k, iz, ix]; asserts nothing non-zero is left over."""
    A = sp.csr_matrix(A)
    N = nz * nx
    iz, ix = np.mgrid[0:nz, 0:nx]
    C = np.zeros((9, nz, nx), dtype=np.complex128)
    for k, (dz, dx) in enumerate(ho.SLOT_OFFSETS):
        jz, jx = iz + dz, ix + dx
        ok = (jz >= 0) & (jz < nz) & (jx >= 0) & (jx < nx)
        rows = (iz * nx + ix)[ok]
        cols = (jz * nx + jx)[ok]
        C[k][ok] = np.asarray(A[rows, cols]).ravel()
    left = A - ho.coefficients_to_csr(C)
    assert left.nnz == 0 or abs(left).max() == 0.0, 'reference matrix has entries outside the 9-point pattern'
    return C


def rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def hetero(nz, nx, seed, complex_c=True):
    rng = np.random.default_rng(seed)
    c = 1800. + 2200. * rng.random((nz, nx))
    if complex_c:
        c = c * (1 + 0.5j / (40. + 60. * rng.random((nz, nx))))     # c + 0.5j c / Q
    rho = 1000. + 600. * rng.random((nz, nx))
    return c, rho


def g1_minizephyr_planes():
    nz, nx = 24, 28
    c, rho = hetero(nz, nx, 11)
    base = dict(nx=nx, nz=nz, dx=7., dz=9., c=c, rho=rho, freq=13., tau=0.4, ky=0.001, nPML=5)
    out = dict(nz=nz, nx=nx, c=c, rho=rho, dx=7., dz=9., freq=13., tau=0.4, ky=0.001, nPML=5)
    sums = []
    for code in range(16):
        fs = tuple(bool(code >> b & 1) for b in range(4))
        sc = dict(base, freeSurf=fs)
        C = planes_from_matrix(zb.MiniZephyr(sc).A, nz, nx)
        mine = ho.minizephyr_coefficients(nz, nx, c, rho, 13., dx=7., dz=9., tau=0.4, ky=0.001, nPML=5, freeSurf=fs)
        assert rel(mine, C) <= 1e-13, ('MiniZephyr planes', fs, rel(mine, C))
        sums.append([C.sum(), np.abs(C).sum(), (C * np.arange(C.size).reshape(C.shape)).sum()])
        if code in (0, 9, 6):
            out['planes_fs%d' % code] = C
    out['checksums'] = np.array(sums)
    # defaults: scalar c, default rho (Gardner), dx default, nPML default
    sc = dict(nx=30, nz=26, c=2500., freq=40.)
    out['planes_defaults'] = planes_from_matrix(zb.MiniZephyr(sc).A, 26, 30)
    mine = ho.minizephyr_coefficients(26, 30, 2500., ho.gardner_rho(ho.as_field(2500., 26, 30, np.complex128)), 40.)
    assert rel(mine, out['planes_defaults']) <= 1e-13
    np.savez_compressed(os.path.join(OUT, 'g1_minizephyr_planes.npz'), **out)


def g2_eurus_planes():
    nz, nx = 22, 26
    c, rho = hetero(nz, nx, 12)
    rng = np.random.default_rng(13)
    theta = 0.6 * rng.random((nz, nx)) - 0.3
    eps = 0.25 * rng.random((nz, nx))
    delta = 0.15 * rng.random((nz, nx))
    out = dict(nz=nz, nx=nx, c=c, rho=rho, dx=7., dz=9., freq=13., tau=0.4, nPML=6, cPML=800., theta=theta, eps=eps, delta=delta)
    for name, kw in (('iso', {}), ('tti', dict(theta=theta, eps=eps, delta=delta)), ('ell', dict(eps=eps, delta=eps))):
        sc = dict(nx=nx, nz=nz, dx=7., dz=9., c=c, rho=rho, freq=13., tau=0.4, nPML=6, cPML=800.)
        sc.update(kw)
        A = sp.csr_matrix(zb.Eurus(sc).A)
        N = nz * nx
        C4 = np.stack([planes_from_matrix(A[r * N:(r + 1) * N, cc * N:(cc + 1) * N], nz, nx) for r in (0, 1) for cc in (0, 1)])
        mine = ho.eurus_coefficients(nz, nx, c, rho, 13., dx=7., dz=9., tau=0.4, nPML=6, cPML=800., **kw)
        for m in range(4):
            scale = np.abs(C4[m]).max()
            assert scale == 0 and np.abs(mine[m]).max() == 0 or rel(mine[m], C4[m]) <= 1e-13, ('Eurus planes', name, m)
        out['planes_' + name] = C4
    np.savez_compressed(os.path.join(OUT, 'g2_eurus_planes.npz'), **out)


def g3_wavefields():
    out = {}
    # the reference's own test configuration (test_MiniZephyr.py:81-114, test_Eurus.py:39-94)
    nx, nz = 100, 200
    sloc = np.array([[25., 25.]])
    rec = (np.arange(5, 196, 10), 60)          # receiver line: iz = 5..195 step 10 at ix = 60
    for name, cls, sc in (
        ('mz', zb.MiniZephyr, dict(c=2500., rho=1., nx=nx, nz=nz, freq=2e2)),
        ('mzhd', zb.MiniZephyrHD, dict(c=2500., rho=1., nx=nx, nz=nz, freq=2e2)),
        ('eu', zb.Eurus, dict(c=2000. * np.ones((nz, nx)), rho=np.ones((nz, nx)), nx=nx, nz=nz, dx=1, dz=1, freq=2e2, nPML=10, cPML=1e3)),
        ('euhd', zb.EurusHD, dict(c=2000. * np.ones((nz, nx)), rho=np.ones((nz, nx)), nx=nx, nz=nz, dx=1, dz=1, freq=2e2, nPML=10, cPML=1e3)),
        ('euell', zb.Eurus, dict(c=2000. * np.ones((nz, nx)), rho=np.ones((nz, nx)), nx=nx, nz=nz, dx=1, dz=1, freq=2e2, nPML=10, cPML=1e3,
                                 theta=np.zeros((nz, nx)), eps=0.2 * np.ones((nz, nx)), delta=0.2 * np.ones((nz, nx)))),
    ):
        op = cls(sc)
        q = (zb.StackedSimpleSource if name.startswith('eu') else zb.SimpleSource)(sc)(sloc)
        u = op * q
        ur = u[:nx * nz, 0].reshape((nz, nx))
        out[name + '_line'] = ur[rec[0], rec[1]]
        # the analytic comparison the reference's tests assert on
        scA = dict(sc)
        if name == 'euell':
            scA.update(eps=0.2, theta=0.)
        scA['c'] = float(np.ravel(sc['c'])[0]); scA['rho'] = 1.
        uA = zb.AnalyticalHelmholtz(scA)(sloc).reshape((nz, nx))
        seg = (uA[40:180, 40:80] - ur[40:180, 40:80]) / abs(uA[40:180, 40:80])
        out[name + '_analytic_err'] = np.sqrt((seg.conj() * seg).sum()) / seg.size
        # oracle check
        if name.startswith('mz'):
            C = ho.minizephyr_coefficients(nz, nx, sc['c'], sc['rho'], 2e2)
            mine = ho.DirectOperator(C, premul=ho.premul_hd(2e2) if name == 'mzhd' else 1.) * q
        else:
            kw = dict(theta=sc['theta'], eps=sc['eps'], delta=sc['delta']) if name == 'euell' else {}
            C = ho.eurus_coefficients(nz, nx, sc['c'], sc['rho'], 2e2, dx=1, dz=1, nPML=10, cPML=1e3, **kw)
            mine = ho.DirectOperator(C, premul=ho.premul_hd(2e2) if name == 'euhd' else 1., eurus=True) * q
        assert np.linalg.norm(mine - u) / np.linalg.norm(u) <= 1e-10, (name, np.linalg.norm(mine - u) / np.linalg.norm(u))
    out['rec_iz'] = rec[0]
    out['rec_ix'] = rec[1]
    # heterogeneous full fields at 64 x 64 (exercises the Eurus z-flip)
    nz = nx = 64
    c, rho = hetero(nz, nx, 21)
    sc = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, rho=rho, freq=12., nPML=8)
    locs = np.array([[300., 320.], [150., 400.]])
    q = zb.SimpleSource(sc)(locs)
    out['het_c'], out['het_rho'], out['het_locs'], out['het_q'] = c, rho, locs, q
    out['het_mz'] = zb.MiniZephyr(sc) * q
    out['het_eu'] = zb.Eurus(sc) * q
    out['het_eu_stacked'] = zb.Eurus(sc) * zb.StackedSimpleSource(sc)(locs)
    C = ho.minizephyr_coefficients(nz, nx, c, rho, 12., dx=10., dz=10., nPML=8)
    assert np.linalg.norm(ho.DirectOperator(C) * q - out['het_mz']) / np.linalg.norm(out['het_mz']) <= 1e-10
    C4 = ho.eurus_coefficients(nz, nx, c, rho, 12., dx=10., dz=10., nPML=8)
    assert np.linalg.norm(ho.DirectOperator(C4, eurus=True) * q - out['het_eu']) / np.linalg.norm(out['het_eu']) <= 1e-10
    # config 1 of BASELINE.json: MiniZephyr 128^2, c=2000, 5 Hz, 1 source at (640, 640) m
    sc1 = dict(nx=128, nz=128, dx=10., dz=10., c=2000., rho=1., nPML=10, freq=5.)
    q1 = zb.SimpleSource(sc1)(np.array([[640., 640.]]))
    u1 = zb.MiniZephyr(sc1) * q1
    out['cfg1_u'] = u1[:, 0]
    C = ho.minizephyr_coefficients(128, 128, 2000., 1., 5., dx=10., dz=10.)
    assert np.linalg.norm((ho.DirectOperator(C) * q1)[:, 0] - u1[:, 0]) / np.linalg.norm(u1) <= 1e-12
    np.savez_compressed(os.path.join(OUT, 'g3_wavefields.npz'), **out)


def g4_multifreq():
    nz, nx = 40, 50
    c, rho = hetero(nz, nx, 31, complex_c=False)
    freqs = [8., 12., 20.]
    sc = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, rho=rho, nPML=6, freqs=freqs, Disc=zb.MiniZephyr, parallel=False, scaleTerm=0.5 - 0.25j)
    locs = np.array([[200., 150.], [310., 220.]])
    q = zb.SimpleSource(sc)(locs)
    out = dict(c=c, rho=rho, freqs=np.array(freqs), q=q, locs=locs)
    mf = zb.MultiFreq(sc)
    out['shared'] = np.stack(list(mf * q))
    qlist = [q * (1 + i) for i in range(3)]
    out['list'] = np.stack(list(mf * qlist))
    out['gen'] = np.stack(list(mf * (qq for qq in qlist)))
    out['onedim'] = np.stack(list(mf * q[:, 0]))
    # ViscoMultiFreq (complex c from Q, dispersion)
    Q = 50. + 100. * np.random.default_rng(32).random((nz, nx))
    scv = dict(sc, c=c, Q=Q, freqBase=10., Disc=zb.MiniZephyr)
    scv.pop('scaleTerm')
    vm = zb.ViscoMultiFreq(scv)
    out['Q'] = Q
    out['visco'] = np.stack(list(vm * q))
    out['visco_c'] = np.stack([np.asarray(s.c).reshape((nz, nx)) for s in vm.subProblems])
    np.savez_compressed(os.path.join(OUT, 'g4_multifreq.npz'), **out)


def g5_sources():
    out = {}
    sc = dict(nx=100, nz=100, dx=1., dz=1.)
    locs = np.array([[50., 50.], [25.5, 30.25], [2.2, 50.1], [97.6, 96.3], [50.3, 1.4]])
    out['locs'] = locs
    for name, cfg in (('nofs', dict(sc)), ('ireg2', dict(sc, ireg=2)), ('ireg0', dict(sc, ireg=0)),
                      ('scaled', dict(nx=100, nz=100, dx=12.5, dz=10., xorig=-100., zorig=50.))):
        use = locs if name != 'scaled' else locs * np.array([12.5, 10.]) + np.array([-100., 50.])
        q = zb.SparseKaiserSource(cfg)(use).tocoo()
        order = np.lexsort((q.row, q.col))
        out[name + '_row'], out[name + '_col'], out[name + '_val'] = q.row[order], q.col[order], q.data[order]
        out[name + '_locs'] = use
    # free-surface mirroring: a location near the top edge with freeSurf[2] (only case that is shape-consistent in the reference)
    cfgfs = dict(sc, freeSurf=(False, False, True, False))
    lfs = np.array([[50.3, 1.4], [20.2, 2.6]])
    q = zb.SparseKaiserSource(cfgfs)(lfs).tocoo()
    order = np.lexsort((q.row, q.col))
    out['fs_row'], out['fs_col'], out['fs_val'], out['fs_locs'] = q.row[order], q.col[order], q.data[order], lfs
    out['simple_idx'] = zb.SimpleSource(sc).linIndexOf(locs)
    out['simple_q_nz'] = np.argwhere(zb.SimpleSource(sc)(locs) != 0)
    out['stacked_shape'] = np.array(zb.StackedSimpleSource(sc)(locs).shape)
    np.savez_compressed(os.path.join(OUT, 'g5_sources.npz'), **out)


def g6_survey():
    from zephyr.middleware import Helm2DProblem, Helm2DSurvey
    nz, nx = 60, 80
    rng = np.random.default_rng(41)
    c = 2000. + 1500. * rng.random((nz, nx))
    rho = 1000. + 200. * rng.random((nz, nx))
    src = np.stack([np.linspace(100, 700, 13), np.full(13, 80.)], 1)
    rec = np.stack([np.linspace(60, 740, 11), np.full(11, 520.)], 1)
    sterms = np.array([1.0 + 0.5j, 0.7 - 0.2j, 1.3 + 0.1j])
    sc = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, rho=rho, nPML=6, freqs=[6., 9., 14.], Disc=zb.MiniZephyrHD, parallel=False,
              sterms=sterms, geom=dict(src=src, rec=rec, mode='fixed'))
    prob, surv = Helm2DProblem(sc), Helm2DSurvey(sc)
    prob.pair(surv)
    d = surv.dpred()
    resid = (rng.standard_normal(d.shape) + 1j * rng.standard_normal(d.shape)) * np.abs(d).mean()
    g_mux = prob.Jtvec(None, resid)
    uF = [np.asarray(x) for x in prob.lazyFields()]
    g_u = prob.Jtvec(None, resid, u=uF)
    out = dict(c=c, rho=rho, src=src, rec=rec, sterms=sterms, freqs=np.array([6., 9., 14.]), dpred=d, resid=resid, g_mux=g_mux, g_u=g_u,
               uF_f1_src3=uF[1][:, 3])
    # relative-geometry survey (receivers move with the source)
    sc2 = dict(sc, geom=dict(src=src, rec=np.stack([np.linspace(-40, 40, 5), np.full(5, 300.)], 1), mode='relative'))
    prob2, surv2 = Helm2DProblem(sc2), Helm2DSurvey(sc2)
    prob2.pair(surv2)
    out['dpred_relative'] = surv2.dpred()
    out['rec_relative'] = sc2['geom']['rec']
    np.savez_compressed(os.path.join(OUT, 'g6_survey.npz'), **out)


def g8_25d():
    '''MiniZephyr25D: the reference's test configurations (test_MiniZephyr.py:35-56,116-152).'''
    nx, nz = 100, 200
    out = {}
    sc = dict(c=2500., rho=1., nx=nx, nz=nz, freq=2e2, nky=20, parallel=False)
    sloc = np.array([[25., 25.]])
    op = zb.MiniZephyr25D(sc)
    u = (op * zb.SimpleSource(sc)(sloc))[:, 0].reshape((nz, nx))
    out['pkys'] = np.array([complex(k).real for k in op.pkys])
    out['premuls'] = np.array([complex(s['premul']).real for s in op.spUpdates])
    out['line'] = u[np.arange(5, 196, 10), 60]
    uA = zb.AnalyticalHelmholtz(dict(sc, **{'3D': True}))(sloc).reshape((nz, nx))
    seg = (uA[40:180, 40:80] - u[40:180, 40:80]) / abs(uA[40:180, 40:80])
    out['analytic_err'] = np.sqrt((seg.conj() * seg).sum()) / seg.size
    sc4 = dict(sc, nky=4)
    u4 = (zb.MiniZephyr25D(sc4) * zb.SimpleSource(sc4)(np.array([[50., 100.]])))[:, 0].reshape((nz, nx))
    out['nky4_line'] = u4[np.arange(5, 196, 10), 60]
    np.savez_compressed(os.path.join(OUT, 'g8_25d.npz'), **out)


def g11_25d_survey():
    """The 2.5-D pairing classes (zephyr/middleware/problem.py:225-238, survey.py:343-346).  In the reference itself
    `Helm25DProblem({'Disc': MiniZephyr25D, ...})` cannot run: MultiFreq hands its own 'Disc' key down (distributors.py:254 masks only
    'freqs'), MiniZephyr25D takes it as the discretisation of its ky sub-problems (minizephyr.py:353-370) and the nested MiniZephyr25D then
    misses the masked 'nky' -- asserted below.  What the pairing is meant to compute is pinned from the reference's own parts instead:
    the fields of `MiniZephyr25D(dict(sc, freq=f)) * q_f` with the sources of `Helm25DSurvey.getSources()`, projected by
    `Helm25DSurvey._lazyProjectFields` (survey.py:152-160)."""
    from zephyr.middleware import Helm25DProblem, Helm25DSurvey
    nz, nx = 48, 64
    rng = np.random.default_rng(77)
    c = 2200. + 900. * rng.random((nz, nx))
    rho = 1000. + 150. * rng.random((nz, nx))
    src = np.stack([np.linspace(120, 500, 5), np.full(5, 90.)], 1)
    rec = np.stack([np.linspace(80, 560, 7), np.full(7, 380.)], 1)
    sterms = np.array([1.0 + 0.25j, 0.8 - 0.1j])
    freqs = [7., 11.]
    sc = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, rho=rho, nPML=6, freqs=freqs, Disc=zb.MiniZephyr25D, nky=6, parallel=False,
              sterms=sterms, geom=dict(src=src, rec=rec, mode='fixed'))
    prob, surv = Helm25DProblem(sc), Helm25DSurvey(sc)
    prob.pair(surv)
    try:
        surv.dpred()
        raise AssertionError('the reference pairing ran: regenerate g11 from it directly')
    except ValueError as exc:
        assert 'nky' in str(exc), exc
    qs = surv.getSources()
    base = {k: v for k, v in sc.items() if k not in ('freqs', 'Disc')}
    u = [zb.MiniZephyr25D(dict(base, freq=f)) * qs[i] for i, f in enumerate(freqs)]
    d = surv._lazyProjectFields(u).ravel()
    out = dict(c=c, rho=rho, src=src, rec=rec, sterms=sterms, freqs=np.array(freqs), nky=6, dpred=d, u_f0_src2=np.asarray(u[0])[:, 2])
    np.savez_compressed(os.path.join(OUT, 'g11_25d_survey.npz'), **out)


def ibm_to_float(words):
    '''IBM System/360 single precision (big-endian uint32 words) -> float64'''
    w = np.asarray(words, dtype=np.uint64)
    sign = np.where(w >> 31, -1.0, 1.0)
    exponent = ((w >> 24) & 0x7f).astype(np.int64) - 64
    mantissa = (w & 0x00ffffff).astype(np.float64) / float(1 << 24)
    return sign * mantissa * np.power(16.0, exponent)


def g9_xhlayr():
    '''The reference's own heterogeneous fixture: notebooks/Time Comprehensive/xhlayr.vp (SEG-Y, IBM floats,
    100 traces x 200 samples) with the geometry of xhlayr.ini (sources at x=15, receivers at x=85, z=15..185 step 2).
    Stores the decoded model and the reference's receiver data for a subset of sources at 100 Hz.'''
    raw = open(os.path.join(REF, 'notebooks', 'Time Comprehensive', 'xhlayr.vp'), 'rb').read()
    ns = int.from_bytes(raw[3220:3222], 'big'); fmt = int.from_bytes(raw[3224:3226], 'big')
    assert ns == 200 and fmt == 1
    ntr = (len(raw) - 3600) // (240 + 4 * ns)
    traces = []
    for t in range(ntr):
        off = 3600 + t * (240 + 4 * ns) + 240
        traces.append(ibm_to_float(np.frombuffer(raw[off:off + 4 * ns], dtype='>u4')))
    c = np.array(traces).T                      # (nz=200, nx=100)
    nz, nx = c.shape
    zs = np.arange(15., 186., 2.)
    src = np.stack([np.full(zs.size, 15.), zs], 1)[::11]       # 8 of the 86 sources
    rec = np.stack([np.full(zs.size, 85.), zs], 1)
    sc = dict(nx=nx, nz=nz, dx=1., dz=1., c=c, freq=100.)
    op = zb.MiniZephyrHD(sc)
    q = zb.SparseKaiserSource(sc)(src)
    u = op * q
    R = zb.SparseKaiserSource(sc)(rec).T
    data = R * u
    out = dict(c=c.astype(np.float64), src=src, rec=rec, freq=100., data=np.asarray(data), u_src0_col60=u[:, 0].reshape((nz, nx))[:, 60])
    Cm = ho.minizephyr_coefficients(nz, nx, c, ho.gardner_rho(c.astype(complex)), 100.)
    mine = ho.DirectOperator(Cm, premul=ho.premul_hd(100.)) * q
    assert np.linalg.norm(mine - u) / np.linalg.norm(u) <= 1e-10
    np.savez_compressed(os.path.join(OUT, 'g9_xhlayr.npz'), **out)
    print('xhlayr model', c.min(), c.max(), 'data', np.abs(data).max())


def g10_omega():
    """OMEGA project I/O and the `zephyr model` job (db.py:35-66,81-271; util.py:21-157; jobs.py:88-207), and Jvec
    (problem.py:87-122), run through the reference.  The project files xhlayr.ini / xhlayr.vp are data fixtures of the
    reference (notebooks/Time Comprehensive) and are copied next to the vectors."""
    import shutil, tempfile, builtins
    np.float = float                                    # db.py:179 uses the alias numpy 2 removed
    builtins.unicode = str                              # db.py:124
    builtins.xrange = range                             # problem.py:101
    from zephyr.middleware import util as zu, db as zdb
    from zephyr.middleware import Helm2DViscoProblem, Helm2DProblem, Helm2DSurvey
    src_dir = os.path.join(REF, 'notebooks', 'Time Comprehensive')
    fx = os.path.join(OUT, 'xhlayr')
    os.makedirs(fx, exist_ok=True)
    for fn in ('xhlayr.ini', 'xhlayr.vp'):
        shutil.copyfile(os.path.join(src_dir, fn), os.path.join(fx, fn))
        os.chmod(os.path.join(fx, fn), 0o644)
    ini = zu.readini(os.path.join(fx, 'xhlayr.ini'))
    out = {}
    for k, v in ini.items():
        if isinstance(v, (int, float, bool, np.ndarray)):
            out['ini_' + k] = np.asarray(v)
        elif isinstance(v, list) and v and isinstance(v[0], int):
            out['ini_' + k] = np.asarray(v)
    out['ini_strings'] = np.array([ini['datain'], ini['dataout'], ini['we']])
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    try:
        for fn in ('xhlayr.ini', 'xhlayr.vp'):
            shutil.copyfile(os.path.join(fx, fn), os.path.join(tmp, fn))
        os.chdir(tmp)
        ds = zdb.FullwvDatastore('xhlayr')
        sc = ds.systemConfig
        out['sc_c'] = np.asarray(sc['c'], dtype=np.float64)
        out['sc_freqs'] = np.asarray(sc['freqs'])
        out['sc_src'] = sc['geom']['src']; out['sc_rec'] = sc['geom']['rec']
        out['sc_scalars'] = np.array([sc['nx'], sc['nz'], sc['dx'], sc['dz'], sc['xorig'], sc['zorig'], sc['nky'], sc['ireg'], sc['freqBase'], sc['tau']], dtype=np.float64)
        out['sc_freeSurf'] = np.array(sc['freeSurf'])
        # the OmegaJob pipeline (jobs.py:88-127,202-207) on three of the project's frequencies and every 6th source
        fid = [9, 19, 29]
        sc2 = dict(sc, freqs=[float(sc['freqs'][i]) for i in fid], Disc=zb.MiniZephyrHD, parallel=False,
                   geom=dict(src=sc['geom']['src'][::6], rec=sc['geom']['rec'], mode='fixed'))
        prob, surv = Helm2DViscoProblem(sc2), Helm2DSurvey(sc2)
        prob.pair(surv)
        data = surv.dpred()
        data.shape = (surv.nrec, surv.nsrc, surv.nfreq)
        zdb.UtoutWriter(sc2)(data)
        out['job_fid'] = np.array(fid); out['job_src_step'] = 6
        out['job_data'] = data
        out['job_utout'] = np.frombuffer(open('xhlayr.utout', 'rb').read(), dtype=np.uint8)
        # damped variant of the writer: omega + i/tau in column 0
        zdb.UtoutWriter(dict(sc2, tau=0.4))(data, ftype='utdamp')
        out['job_utout_tau'] = np.frombuffer(open('xhlayr.utdamp', 'rb').read(), dtype=np.uint8)
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp)
    # Jvec on the g6 survey (fixed geometry)
    g6 = np.load(os.path.join(OUT, 'g6_survey.npz'))
    sc6 = dict(nx=80, nz=60, dx=10., dz=10., c=g6['c'], rho=g6['rho'], nPML=6, freqs=[6., 9., 14.], Disc=zb.MiniZephyrHD, parallel=False,
               sterms=g6['sterms'], geom=dict(src=g6['src'], rec=g6['rec'], mode='fixed'))
    prob, surv = Helm2DProblem(sc6), Helm2DSurvey(sc6)
    prob.pair(surv)
    rng = np.random.default_rng(77)
    v = rng.standard_normal(60 * 80) * 30.
    out['jvec_v'] = v
    out['jvec'] = prob.Jvec(None, v)
    np.savez_compressed(os.path.join(OUT, 'g10_omega.npz'), **out)
    print('g10: ini nom', ini['nom'], 'utout bytes', out['job_utout'].size, '|Jvec|', np.abs(out['jvec']).max())


def g7_analytic():
    sc = dict(c=2500., rho=1., nx=100, nz=200, freq=2e2)
    out = {}
    for name, s in (('green2d', sc), ('green3d', dict(sc, **{'3D': True})), ('stretch', dict(sc, eps=0.2, theta=0.3, dx=2., dz=1.5))):
        out[name] = zb.AnalyticalHelmholtz(s)(np.array([[25., 25.]]))
    np.savez_compressed(os.path.join(OUT, 'g7_analytic.npz'), **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['g1', 'g2', 'g3', 'g4', 'g5', 'g6', 'g7', 'g8', 'g9', 'g10', 'g11']
    table = dict(g1=g1_minizephyr_planes, g2=g2_eurus_planes, g3=g3_wavefields, g4=g4_multifreq, g5=g5_sources, g6=g6_survey, g7=g7_analytic, g8=g8_25d, g9=g9_xhlayr, g10=g10_omega, g11=g11_25d_survey)
    for name in which:
        table[name]()
        print('wrote', name, flush=True)
    for f in sorted(os.listdir(OUT)):
        print('%8.1f KB  %s' % (os.path.getsize(os.path.join(OUT, f)) / 1024., f))
