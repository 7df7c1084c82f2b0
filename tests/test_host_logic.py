"""CPU: host-side mirror of the reference interface (config layer, sources, analytical oracle,
dispatcher routing) against golden vectors from the reference.  No GPU compute."""
import os
import re
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import helm_oracle as ho
import zephyr_amd as za
from zephyr_amd import config

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def load(name):
    return np.load(os.path.join(GOLD, name))


# ---- config layer ------------------------------------------------------------------------------
def test_initmap_merge_and_required():
    class A(config.AttributeMapper):
        initMap = {'a': (True, None, np.int64), 'b': (False, '_b', np.float64)}

    class B(A):
        initMap = {'c': (False, None, tuple), 'b': (False, '_bb', np.float64)}

    assert set(B.initMap) == {'a', 'b', 'c'}
    obj = B({'a': 3.0, 'b': 2, 'c': [1, 2]})
    assert obj.a == 3 and isinstance(obj.a, np.int64)
    assert obj._bb == 2.0 and obj.c == (1, 2)
    with pytest.raises(ValueError):
        B({'b': 1})


def test_complex_scalar_into_float_field_takes_real_part():
    class A(config.AttributeMapper):
        initMap = {'v': (False, None, np.float64)}
    assert A({'v': 3 + 0j}).v == 3.0


def test_sccache_masks_and_clears():
    class W(config.BaseSCCache):
        initMap = {'x': (True, None, np.int64), 'secret': (False, '_s', None)}
        maskKeys = {'secret'}
        cacheItems = ['_thing']
    w = W({'x': 1, 'secret': 5, 'other': 7})
    assert 'secret' not in w.systemConfig and w.systemConfig['other'] == 7
    w._thing = 4
    w.systemConfig = dict(w.systemConfig)
    assert not hasattr(w, '_thing')
    f = config.SCFilter(W)
    assert f({'x': 1, 'zzz': 2}) == {'x': 1}
    with pytest.raises(ValueError):
        f({'zzz': 2})


def test_discretisation_defaults_match_reference_contract():
    sc = dict(nx=30, nz=20, c=2500., freq=40.)
    d = za.MiniZephyr(sc)
    assert d.dx == 1.0 and d.dz == 1.0 and d.nPML == 10 and d.ky == 0.0 and d.tau == np.inf
    assert d.freeSurf == (False, False, False, False)
    assert d.c.shape == (20, 30) and d.c.dtype == np.complex128
    assert np.allclose(d.rho, 310. * 2500. ** 0.25)            # Gardner default (discretization.py:70)
    assert d.premul == 1.0
    assert d.mord == (30, 1)
    assert za.MiniZephyrHD(sc).premul == np.sqrt(2j * np.pi * 40.)
    e = za.Eurus(sc)
    assert e.mord == (-30, 1) and e.cPML == 1e3 and e.shape == (1200, 1200)
    assert np.all(e.theta == 0) and np.all(e.eps == 0) and np.all(e.delta == 0)
    assert za.EurusHD(sc).premul == np.sqrt(2j * np.pi * 40.)
    assert d.factors is False
    with pytest.raises(ValueError):
        za.MiniZephyr(dict(nx=3, nz=3, freq=1.))                # 'c' required


def test_nondefault_mord_is_refused():
    d = za.MiniZephyr(dict(nx=30, nz=20, c=2500., freq=40., mord=(1, 30)))
    with pytest.raises(NotImplementedError):
        d._assemble_args()


def test_discretisation_is_picklable_without_handle():
    import pickle
    d = za.Eurus(dict(nx=30, nz=20, c=2500., freq=40.))
    d2 = pickle.loads(pickle.dumps(d))
    assert d2.nx == 30 and d2.factors is False


# ---- sources --------------------------------------------------------------------------------------
def coo_sorted(q):
    q = q.tocoo()
    order = np.lexsort((q.row, q.col))
    return q.row[order], q.col[order], q.data[order]


@pytest.mark.parametrize('name,cfg', [
    ('nofs', dict(nx=100, nz=100, dx=1., dz=1.)),
    ('ireg2', dict(nx=100, nz=100, dx=1., dz=1., ireg=2)),
    ('ireg0', dict(nx=100, nz=100, dx=1., dz=1., ireg=0)),
    ('scaled', dict(nx=100, nz=100, dx=12.5, dz=10., xorig=-100., zorig=50.)),
    ('fs', dict(nx=100, nz=100, dx=1., dz=1., freeSurf=(False, False, True, False))),
])
def test_sparse_kaiser_source_matches_reference(name, cfg):
    g = load('g5_sources.npz')
    q = za.SparseKaiserSource(cfg)(g[name + '_locs'])
    row, col, val = coo_sorted(q)
    assert q.shape == (10000, len(g[name + '_locs']))
    assert np.array_equal(row, g[name + '_row']) and np.array_equal(col, g[name + '_col'])
    assert np.allclose(val, g[name + '_val'], rtol=1e-14, atol=1e-18)


def test_simple_and_stacked_sources():
    g = load('g5_sources.npz')
    sc = dict(nx=100, nz=100, dx=1., dz=1.)
    s = za.SimpleSource(sc)
    assert np.array_equal(s.linIndexOf(g['locs']), g['simple_idx'])
    q = s(g['locs'])
    assert np.array_equal(np.argwhere(q != 0), g['simple_q_nz'])
    assert tuple(za.StackedSimpleSource(sc)(g['locs']).shape) == tuple(g['stacked_shape'])


def test_kaiser_on_node_equals_simple():
    """reference test_Sources.py:51-68: on-node Kaiser source with dx=dz=1 equals the delta to 1e-10"""
    sc = dict(nx=100, nz=100)
    locs = np.array([[25., 25.], [50., 50.], [75., 75.], [10., 80.]])
    dense = za.KaiserSource(sc)(locs)
    sparse = za.SparseKaiserSource(sc)(locs)
    assert np.abs(dense - sparse.toarray()).max() == 0            # test_Sources.py:14-32
    assert np.abs(dense - za.SimpleSource(sc)(locs)).max() < 1e-10


# ---- analytical -------------------------------------------------------------------------------------
def test_analytical_matches_reference():
    g = load('g7_analytic.npz')
    sc = dict(c=2500., rho=1., nx=100, nz=200, freq=2e2)
    loc = np.array([[25., 25.]])
    assert np.allclose(za.AnalyticalHelmholtz(sc)(loc), g['green2d'], rtol=1e-13, atol=0)
    assert np.allclose(za.AnalyticalHelmholtz(dict(sc, **{'3D': True}))(loc), g['green3d'], rtol=1e-13, atol=0)
    assert np.allclose(za.AnalyticalHelmholtz(dict(sc, eps=0.2, theta=0.3, dx=2., dz=1.5))(loc), g['stretch'], rtol=1e-13, atol=0)


# ---- dispatchers (with an oracle-backed test double standing in for the GPU Disc) ---------------------
class OracleDisc(za.MiniZephyr):
    """Test double: same config ingestion as the product class, arithmetic by the CPU oracle."""

    def __mul__(self, rhs):
        C = ho.minizephyr_coefficients(int(self.nz), int(self.nx), self.c, self.rho, complex(self.freq), dx=self.dx, dz=self.dz,
                                       nPML=int(self.nPML), tau=self.tau, ky=self.ky, freeSurf=self.freeSurf)
        if sp.issparse(rhs):
            rhs = rhs.toarray()
        return ho.DirectOperator(C, premul=self.premul) * rhs


def test_multifreq_routing_and_order():
    g = load('g4_multifreq.npz')
    nz, nx = g['c'].shape
    sc = dict(nx=nx, nz=nz, dx=10., dz=10., c=g['c'], rho=g['rho'], nPML=6, freqs=list(g['freqs']), Disc=OracleDisc,
              parallel=False, scaleTerm=0.5 - 0.25j)
    mf = za.MultiFreq(sc)
    q = g['q']
    assert [complex(s.freq).real for s in mf.subProblems] == list(g['freqs'])
    out = mf * q
    assert hasattr(out, '__next__')                              # generator, in `freqs` order
    assert np.linalg.norm(np.stack(list(out)) - g['shared']) / np.linalg.norm(g['shared']) < 1e-10
    qlist = [q * (1 + i) for i in range(3)]
    assert np.linalg.norm(np.stack(list(mf * qlist)) - g['list']) / np.linalg.norm(g['list']) < 1e-10
    assert np.linalg.norm(np.stack(list(mf * (qq for qq in qlist))) - g['gen']) / np.linalg.norm(g['gen']) < 1e-10
    one = np.stack(list(mf * q[:, 0]))
    assert one.shape == g['onedim'].shape
    assert np.linalg.norm(one - g['onedim']) / np.linalg.norm(g['onedim']) < 1e-10
    assert 'scaleTerm' not in mf.systemConfig and 'freqs' not in mf.systemConfig


def test_visco_multifreq_velocity_and_result():
    g = load('g4_multifreq.npz')
    nz, nx = g['c'].shape
    sc = dict(nx=nx, nz=nz, dx=10., dz=10., c=g['c'], rho=g['rho'], nPML=6, freqs=list(g['freqs']), Disc=OracleDisc,
              parallel=False, Q=g['Q'], freqBase=10.)
    vm = za.ViscoMultiFreq(sc)
    cs = np.stack([np.asarray(s.c).reshape((nz, nx)) for s in vm.subProblems])
    assert np.allclose(cs, g['visco_c'], rtol=1e-14, atol=0)
    assert np.linalg.norm(np.stack(list(vm * g['q'])) - g['visco']) / np.linalg.norm(g['visco']) < 1e-10


def test_sub_problems_share_one_private_copy_of_the_model():
    """Round 6: a dispatcher makes ONE complex128 copy of the velocity model for all its sub-problems (read-only; config.cast_value hands a read-only array of the
    right type on as it is) -- the reference casts per sub-problem (discretization.py:24-31 through galoshes' initMap).  The caller's array is not aliased: changing it
    afterwards changes nothing a sub-problem holds; a per-frequency model (ViscoMultiFreq) is cast per sub-problem as before."""
    g = load('g4_multifreq.npz')
    nz, nx = g['c'].shape
    c = np.array(g['c'], dtype=np.float64)
    sc = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, rho=g['rho'], nPML=6, freqs=list(g['freqs']), Disc=za.MiniZephyr)
    mf = za.MultiFreq(sc)
    subs = mf.subProblems
    assert all(s.c is subs[0].c for s in subs) and subs[0].c.dtype == np.complex128 and not subs[0].c.flags.writeable
    assert not np.shares_memory(subs[0].c, c)
    before = subs[0].c.copy()
    c += 1.0
    assert np.array_equal(subs[0].c, before)
    vm = za.ViscoMultiFreq(dict(sc, Q=g['Q'], freqBase=10.))
    vs = vm.subProblems
    assert vs[0].c is not vs[1].c
    w = np.ones((nz, nx), dtype=np.complex128)
    w.setflags(write=False)
    assert config.cast_value(np.complex128, w) is w                          # (as numpy's own cast does for an array of that type)
    f = np.ones((nz, nx)); f.setflags(write=False)
    assert config.cast_value(np.complex128, f) is not f and config.cast_value(np.complex128, f).dtype == np.complex128


def test_wrapper_factors_flag():
    sc = dict(nx=30, nz=20, c=2500., freqs=[5., 6.], Disc=za.MiniZephyr)
    mf = za.MultiFreq(sc)
    assert mf.factors is False
    assert len(mf.subProblems) == 2
    assert mf.factors is False
    del mf.factors


# ---- models ------------------------------------------------------------------------------------------
def test_marmousi_like_is_deterministic_and_in_range():
    from zephyr_amd.models import marmousi_like
    a = marmousi_like(96, 128, 10.)
    b = marmousi_like(96, 128, 10.)
    assert np.array_equal(a, b)
    assert a.shape == (96, 128) and a.min() >= 1500. and a.max() <= 5500.
    assert a.std() > 300.


def test_minizephyr25d_wavenumbers_and_weights():
    g = load('g8_25d.npz')
    op = za.MiniZephyr25D(dict(c=2500., rho=1., nx=100, nz=200, freq=2e2, nky=20, parallel=False))
    assert np.allclose(op.pkys, g['pkys'], rtol=1e-14, atol=0)
    assert np.allclose([complex(u['premul']).real for u in op.spUpdates], g['premuls'], rtol=1e-14, atol=0)
    assert len(op.subProblems) == 20 and op.subProblems[0].__class__ is za.MiniZephyr
    assert np.isclose(op.scaleTerm, np.exp(1j * np.pi) / (4 * np.pi))
    assert 'nky' not in op.systemConfig
