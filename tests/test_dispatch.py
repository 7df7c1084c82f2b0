"""CPU: the in-process dispatcher (zephyr_amd.dispatch) behind MultiFreq's parallel mode -- the counterpart of the
reference's multiprocessing.Pool in BaseMPDist.__mul__ (zephyr/backend/distributors.py:127-173).  A test double stands in
for the GPU operator: same config ingestion, arithmetic by the CPU oracle, and it records which thread / "device" ran what."""
import os
import threading
import time

import numpy as np
import pytest
import scipy.sparse as sp

import zephyr_amd as za
from zephyr_amd import dispatch
from oracle import helm_oracle as ho

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
LOG = []
LOCK = threading.Lock()


class RecordingDisc(za.MiniZephyr):

    def reserve(self, nrhs, rows=None, concurrent=1):
        with LOCK:
            LOG.append(('reserve', int(nrhs), self.device, int(concurrent)))

    def prefactor(self):
        with LOCK:
            LOG.append(('prefactor', complex(self.freq).real, self.device, threading.current_thread().name))

    def __mul__(self, rhs):
        with LOCK:
            LOG.append(('solve', complex(self.freq).real, self.device, threading.current_thread().name))
        C = ho.minizephyr_coefficients(int(self.nz), int(self.nx), self.c, self.rho, complex(self.freq), dx=self.dx, dz=self.dz,
                                       nPML=int(self.nPML), tau=self.tau, ky=self.ky, freeSurf=self.freeSurf)
        if sp.issparse(rhs):
            rhs = rhs.toarray()
        return ho.DirectOperator(C, premul=self.premul) * rhs


class FailingDisc(RecordingDisc):

    def __mul__(self, rhs):
        if abs(complex(self.freq).real - 6.) < 1e-12:
            raise ArithmeticError('frequency 6 does not converge')
        return RecordingDisc.__mul__(self, rhs)


def config(**extra):
    g = np.load(os.path.join(GOLD, 'g4_multifreq.npz'))
    nz, nx = g['c'].shape
    sc = dict(nx=nx, nz=nz, dx=10., dz=10., c=g['c'], rho=g['rho'], nPML=6, freqs=list(g['freqs']), Disc=RecordingDisc)
    sc.update(extra)
    return g, sc


def test_parallel_mode_matches_serial_bit_for_bit_and_keeps_order(monkeypatch):
    monkeypatch.setenv('HELM_DEVICES', '0,1')
    g, sc = config(scaleTerm=0.5 - 0.25j)
    serial = list(za.MultiFreq(dict(sc, parallel=False)) * g['q'])
    del LOG[:]
    mf = za.MultiFreq(sc)
    assert mf.parallel and mf.nWorkers == 2 and mf.devices == [0, 1]
    out = mf * g['q']
    assert hasattr(out, '__next__')
    par = list(out)
    assert len(par) == len(serial) == 3
    for a, b in zip(par, serial):
        assert np.array_equal(a, b)                              # same arithmetic, same order of the results
    assert np.linalg.norm(np.stack(par) - g['shared']) / np.linalg.norm(g['shared']) < 1e-10
    # frequency-major dealing: frequencies 0 and 2 on device 0, frequency 1 on device 1; every solve was prefactored first,
    # on the prepare thread of its device
    freqs = list(g['freqs'])
    assert [s.device for s in mf.subProblems] == [0, 1, 0]
    # before the workers start, each GPU's scratch is booked once for the workers that will run on it (two on GPU 0, one on GPU 1)
    assert sorted(e for e in LOG if e[0] == 'reserve') == [('reserve', g['q'].shape[1], 0, 2), ('reserve', g['q'].shape[1], 1, 1)]
    for f, d in zip(freqs, (0, 1, 0)):
        ev = [e for e in LOG if e[1] == f and e[0] != 'reserve']
        assert [e[0] for e in ev] == ['prefactor', 'solve'] and all(e[2] == d for e in ev)
        assert ev[0][3] == 'helm-prep%d' % d and ev[1][3] == 'helm-solve%d' % d
    # list and generator right-hand sides are consumed in order at submission, like the reference's apply_async loop
    qlist = [g['q'] * (1 + i) for i in range(3)]
    assert np.linalg.norm(np.stack(list(mf * qlist)) - g['list']) / np.linalg.norm(g['list']) < 1e-10
    assert np.linalg.norm(np.stack(list(mf * (qq for qq in qlist))) - g['gen']) / np.linalg.norm(g['gen']) < 1e-10
    del mf.factors


def test_fewer_frequencies_than_devices_splits_the_sources(monkeypatch):
    monkeypatch.setenv('HELM_DEVICES', '0,1,2,3')
    g, sc = config()
    sc['freqs'] = sc['freqs'][:2]
    ref = list(za.MultiFreq(dict(sc, parallel=False)) * g['q'])
    del LOG[:]
    mf = za.MultiFreq(sc)
    out = list(mf * g['q'])
    assert np.allclose(np.stack(out), np.stack(ref), rtol=1e-13, atol=0)
    solves = sorted((e[1], e[2]) for e in LOG if e[0] == 'solve')
    f0, f1 = sc['freqs']
    assert solves == [(f0, 0), (f0, 1), (f1, 2), (f1, 3)]        # two GPUs per frequency, each with half of the sources
    assert mf.factors is False                                   # (doubles hold no device handle)
    # nWorkers caps the GPUs used (reference: pool size, distributors.py:92-96)
    mf2 = za.MultiFreq(dict(sc, nWorkers=1))
    assert mf2.devices == [0] and mf2.nWorkers == 1


def test_failure_surfaces_at_the_failing_frequency(monkeypatch):
    monkeypatch.setenv('HELM_DEVICES', '0,1')
    g, sc = config(Disc=FailingDisc)
    sc['freqs'] = [5., 6., 7.]
    it = za.MultiFreq(sc) * g['q']
    assert next(it).shape == g['q'].shape
    with pytest.raises(ArithmeticError):
        next(it)


def test_pipeline_overlaps_prepare_with_solve():
    'the prepare step of item k+1 runs while item k is being solved, and never more than `lookahead` items ahead'
    log = []

    def make(k):
        def prep():
            log.append(('p0', k, time.perf_counter()))
            time.sleep(0.05)
            return k

        def solve(p):
            assert p == k
            log.append(('s0', k, time.perf_counter()))
            time.sleep(0.1)
            log.append(('s1', k, time.perf_counter()))
            return k * k
        return dispatch.WorkItem(solve, prep)
    t0 = time.perf_counter()
    res = list(dispatch.pipelined([make(k) for k in range(5)], device=0, lookahead=1))
    wall = time.perf_counter() - t0
    assert res == [0, 1, 4, 9, 16]
    assert wall < 5 * 0.15 - 0.1                                   # serial would take 0.75 s
    t = dict(((a, k), v) for a, k, v in log)
    for k in range(1, 5):
        assert t[('p0', k)] < t[('s1', k - 1)]                     # prepared during the previous solve
    for k in range(3, 5):
        assert t[('p0', k)] >= t[('s0', k - 2)] - 1e-3             # but not before item k-2 has left the queue


def test_single_device_under_a_launcher(monkeypatch):
    monkeypatch.delenv('HELM_DEVICES', raising=False)
    monkeypatch.setenv('LOCAL_RANK', '3')
    assert dispatch.visible_devices() == [3]


def test_problem_work_items_cover_every_source_once(monkeypatch):
    """HelmBaseProblem._deviceItems: (frequency, source-batch) items for the device-resident dpred / Jtvec loops -- frequency-major over the
    GPUs, sources split when there are spare ones; every (frequency, source) pair exactly once, replicas only on GPUs other than the
    frequency's own."""
    from zephyr_amd.problem import Helm2DProblem
    from zephyr_amd.survey import Helm2DSurvey
    g = np.load(os.path.join(GOLD, 'g6_survey.npz'))
    nz, nx = g['c'].shape
    sc = dict(nx=nx, nz=nz, dx=10., dz=10., c=g['c'], rho=g['rho'], nPML=6, freqs=list(g['freqs']), Disc=RecordingDisc,
              sterms=g['sterms'], geom=dict(src=g['src'], rec=g['rec'], mode='fixed'))
    for devices, nfreq_expected_split in (('0', 1), ('0,1', 1), ('0,1,2,3,4,5,6,7', 2)):
        monkeypatch.setenv('HELM_DEVICES', devices)
        prob, surv = Helm2DProblem(sc), Helm2DSurvey(sc)
        prob.pair(surv)
        nsrc, nfreq = surv.nsrc, surv.nfreq
        devs, items = prob._deviceItems(list(range(nfreq)), nsrc)
        assert devs == [int(d) for d in devices.split(',')]
        seen = np.zeros((nfreq, nsrc), int)
        per_freq = {}
        for w, op, ifreq, c0, c1 in items:
            assert 0 <= w < len(devs) and op.device == devs[w]
            seen[ifreq, c0:c1] += 1
            per_freq.setdefault(ifreq, []).append(op)
        assert np.all(seen == 1)
        assert all(len(ops) == nfreq_expected_split for ops in per_freq.values())
        for ifreq, ops in per_freq.items():
            assert ops[0] is prob.system.subProblems[ifreq]
            assert len(set(o.device for o in ops)) == len(ops)
        # a rank that owns a subset of the frequencies (one process per GPU) keeps them on its own GPU
        devs1, items1 = prob._deviceItems([1], nsrc)
        assert sorted((c0, c1) for _, _, i, c0, c1 in items1 if i == 1)[0][0] == 0 and sum(c1 - c0 for _, _, _, c0, c1 in items1) == nsrc


def test_strict_pipeline_prepares_exactly_one_item_ahead():
    'strict=True: the preparation of item k+1 starts when the solve of item k starts, never earlier (3-D operators: heavy on the GPU and in memory)'
    log = []

    def make(k):
        def prep():
            log.append(('p0', k, time.perf_counter()))
            time.sleep(0.02)
            return k

        def solve(p):
            log.append(('s0', k, time.perf_counter()))
            time.sleep(0.08)
            log.append(('s1', k, time.perf_counter()))
            return k
        return dispatch.WorkItem(solve, prep)
    assert list(dispatch.pipelined([make(k) for k in range(4)], device=0, lookahead=1, strict=True)) == [0, 1, 2, 3]
    t = dict(((a, k), v) for a, k, v in log)
    for k in range(1, 4):
        assert t[('p0', k)] >= t[('s0', k - 1)] - 1e-3            # not before the previous item's solve has started
        assert t[('p0', k)] < t[('s1', k - 1)]                    # but during it


def test_a_result_object_that_is_never_iterated_starts_no_worker(monkeypatch):
    """the pipelines start with the first result that is asked for, so dropping `sys * q` un-iterated leaves no thread spinning
    in a throttle with multi-GB results in its hands."""
    monkeypatch.setenv('HELM_DEVICES', '0,1')
    g, sc = config()
    del LOG[:]
    mf = za.MultiFreq(sc)
    before = threading.active_count()
    res = mf * g['q']
    time.sleep(0.2)
    assert threading.active_count() == before and not [e for e in LOG if e[0] in ('solve', 'prefactor', 'reserve')]
    del res
    assert not mf.__dict__.get('_pipes')
    out = list(mf * g['q'])                           # and the next call is a normal one
    assert len(out) == len(sc['freqs'])


def test_a_second_call_stops_the_workers_of_an_undrained_first_one(monkeypatch):
    """two sets of threads must never drive the same operator handles.  The first call's generator is advanced by one result and
    kept; the second call joins the first's pipelines before it creates its own."""
    monkeypatch.setenv('HELM_DEVICES', '0,1')
    g, sc = config()
    serial = list(za.MultiFreq(dict(sc, parallel=False)) * g['q'])
    mf = za.MultiFreq(sc)
    first = mf * g['q']
    u0 = next(first)
    assert np.array_equal(u0, serial[0])
    old_threads = [t for p in mf.__dict__['_pipes'] for t in p._threads]
    assert old_threads
    second = mf * g['q']
    assert not any(t.is_alive() for t in old_threads)          # joined by the second call, before it has started anything of its own
    second = list(second)
    for a, b in zip(second, serial):
        assert np.array_equal(a, b)
    del first


def test_an_unstarted_result_superseded_by_a_later_call_refuses_to_start(monkeypatch):
    """`g1 = A * r1; g2 = A * r2` and then iterating both must not start two sets of pipelines on the same (non-thread-safe)
    handles: the second product owns them, the first one's token is stale and its first next() says so instead of starting workers."""
    monkeypatch.setenv('HELM_DEVICES', '0,1')
    g, sc = config()
    serial = list(za.MultiFreq(dict(sc, parallel=False)) * g['q'])
    mf = za.MultiFreq(sc)
    first = mf * g['q']
    second = mf * g['q']
    before = threading.active_count()
    with pytest.raises(RuntimeError, match='superseded'):
        next(first)
    assert threading.active_count() == before                  # nothing was started on behalf of the stale call
    out = list(second)
    for a, b in zip(out, serial):
        assert np.array_equal(a, b)


def test_a_prepare_step_that_breaks_outside_its_own_guard_does_not_hang_the_pipeline(monkeypatch):
    """The prepare thread hands every item to the solve thread, whatever happened to it: an exception raised around `_run_prepare` (a broken subclass, an
    instrumented method -- it did happen with a profiling wrapper) surfaces where the item's result is consumed instead of leaving the solve thread waiting
    on an empty queue."""
    from zephyr_amd import dispatch

    def broken(item):
        raise RuntimeError('prepare wrapper broke')
    monkeypatch.setattr(dispatch.DevicePipeline, '_run_prepare', staticmethod(broken))
    items = [dispatch.WorkItem((lambda prepared: 1), (lambda: None)) for _ in range(3)]
    out = dispatch.pipelined(items, device=0, lookahead=1)
    import pytest
    with pytest.raises(RuntimeError):
        next(out)


def test_blas_pools_are_single_threaded_while_a_pipeline_runs_and_restored_after():
    """A threaded BLAS call beside a running pipeline leaves its workers spinning, and under a container CPU quota the scheduler then freezes the whole
    process (problem.py, _norm2): the dispatcher holds the pools to one thread from start() until its last thread has finished, then gives the caller's
    limits back."""
    tpc = pytest.importorskip('threadpoolctl')
    from zephyr_amd import dispatch

    def blas_threads():
        return [p['num_threads'] for p in tpc.threadpool_info() if p.get('user_api') == 'blas']
    before = blas_threads()
    if not before or max(before) < 2:
        pytest.skip('no multi-threaded BLAS pool in this process')
    seen = []
    items = [dispatch.WorkItem((lambda _p: seen.append(blas_threads())), None) for _ in range(3)]
    assert list(dispatch.pipelined(items, device=0, lookahead=1)) == [None] * 3
    assert all(v == [1] * len(v) for v in seen), seen
    assert blas_threads() == before


def test_grouped_prepare_takes_items_in_sets_and_an_error_of_the_set_reaches_each_of_its_items():
    """group > 1: every item's own prepare() runs, then ONE group_prepare() for the set (discretization.prefactor_many: the factorisations of the set's operators
    in the same launches); the first item goes alone so that the solve thread starts early; results keep their order; an exception of the set's step surfaces
    at every item of that set and at no other.  (The reference hands its pool one sub-problem at a time: distributors.py:161-168.)"""
    from zephyr_amd import dispatch
    seen = []
    items = [dispatch.WorkItem((lambda p: 10 * p), (lambda k=k: k)) for k in range(7)]
    assert list(dispatch.pipelined(items, device=0, lookahead=1, group=2, group_prepare=lambda ps: seen.append(list(ps)))) == [0, 10, 20, 30, 40, 50, 60]
    assert seen == [[0], [1, 2], [3, 4], [5, 6]]

    def boom(ps):
        if 3 in ps:
            raise RuntimeError('set with item 3 failed')
    items = [dispatch.WorkItem((lambda p: p), (lambda k=k: k)) for k in range(6)]
    futs = [it.future for it in items]
    pipe = dispatch.DevicePipeline(0, 1, False, 1, 2, boom)
    pipe.start(items)
    got = []
    for f in futs:
        try:
            got.append(f.result())
        except RuntimeError as exc:
            got.append(str(exc))
    pipe.join()
    assert got == [0, 1, 2, 'set with item 3 failed', 'set with item 3 failed', 5]          # sets: [0], [1, 2], [3, 4], [5]
    # an item whose own prepare fails is left out of the set's step and fails alone
    def prep(k):
        if k == 2:
            raise ValueError('item 2')
        return k
    sets = []
    items = [dispatch.WorkItem((lambda p: p), (lambda k=k: prep(k))) for k in range(5)]
    futs = [it.future for it in items]
    pipe = dispatch.DevicePipeline(0, 1, False, 1, 2, lambda ps: sets.append(list(ps)))
    pipe.start(items)
    res = []
    for f in futs:
        try:
            res.append(f.result())
        except ValueError as exc:
            res.append(str(exc))
    pipe.join()
    assert res == [0, 1, 'item 2', 3, 4] and sets == [[0], [1], [3, 4]]
