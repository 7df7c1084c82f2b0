#!/usr/bin/env python3
"""Where the host time of config 4 (dpred + Jtvec on 512^2, 8 freqs x 64 sources) goes: cProfile of one dpred(m) and one Jtvec(m, v) after a warm-up."""
import cProfile, pstats, io, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.build()
import zephyr_amd as za
from zephyr_amd.models import marmousi_like, box_smooth
from zephyr_amd.problem import Helm2DProblem
from zephyr_amd.survey import Helm2DSurvey
n, dx, nf, ns, nr = 512, 10.0, 8, 64, 128
ctrue = marmousi_like(n, n, dx); ccur = box_smooth(ctrue, 12)
freqs = list(np.linspace(3.0, 10.0, nf))
src = np.stack([np.linspace(200.0, 4920.0, ns), np.full(ns, 20.0)], axis=1)
rec = np.stack([np.linspace(100.0, dx * n - 100.0, nr), np.full(nr, 20.0)], axis=1)
sc = dict(nx=n, nz=n, dx=dx, dz=dx, freqs=freqs, Disc=za.Eurus, geom=dict(src=src, rec=rec, mode='fixed'), batch=ns, c=ctrue)
p, sv = Helm2DProblem(sc), Helm2DSurvey(sc); p.pair(sv)
dobs = sv.dpred()
m = ccur.ravel()
d = sv.dpred(m); resid = d - dobs
gq = p.Jtvec(m, resid)
# the solves run on the dispatcher's threads: wall-clock timers around the calls they make
import collections, functools, threading
ACC = collections.defaultdict(lambda: [0, 0.0]); LK = threading.Lock()
LOG = []; T00 = [0.0]
def timed(cls, name):
    f = getattr(cls, name)
    @functools.wraps(f)
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            with LK:
                e = ACC[cls.__name__ + '.' + name]; e[0] += 1; e[1] += time.perf_counter() - t0
    setattr(cls, name, staticmethod(w) if isinstance(cls.__dict__.get(name), staticmethod) else w)
import gc
_gc_t = [0.0]
def _gc_cb(phase, info):
    if phase == 'start': _gc_t[0] = time.perf_counter()
    else:
        with LK: LOG.append((1e3 * (_gc_t[0] - T00[0]), 1e3 * (time.perf_counter() - _gc_t[0]), threading.current_thread().name, 'PYTHON GC generation %d (collected %d)' % (info['generation'], info['collected'])))
gc.callbacks.append(_gc_cb)
if os.environ.get('C4_GC') == 'freeze': gc.collect(); gc.freeze()
if os.environ.get('C4_GC') == 'off': gc.disable()
from zephyr_amd import dispatch, discretization
for nm in ('__init__', 'prefactor', 'rhsFromSparseDevice', 'solveDevice', 'sampleDevice', 'imagingAccumulateDevice', 'prepare', '_ensure_handle', '_assemble'):
    for cls in (za.Eurus, discretization.BaseDiscretization):
        if nm in cls.__dict__:
            timed(cls, nm)
from zephyr_amd import _lib as _zl
_L = _zl.load()
def timed_c(name):
    f = getattr(_L, name)
    def w(*a):
        t0 = time.perf_counter()
        try:
            return f(*a)
        finally:
            with LK:
                e = ACC['libhelm.' + name]; e[0] += 1; e[1] += time.perf_counter() - t0
                LOG.append((1e3 * (t0 - T00[0]), 1e3 * (time.perf_counter() - t0), threading.current_thread().name, name))
    setattr(_L, name, w)
for nm in ('helm_create', 'helm_set_model', 'helm_assemble', 'helm_prefactor_n', 'helm_solve_device', 'helm_rhs_from_coo_device_layout', 'helm_sample_device',
           'helm_imaging_accumulate_device', 'helm_destroy', 'helm_set_tolerance_hint'):
    timed_c(nm)
timed(dispatch.DevicePipeline, '_run_prepare'); timed(dispatch.DevicePipeline, '_run_solve')
timed(Helm2DSurvey, 'getResidualSources'); timed(Helm2DSurvey, 'getSources'); timed(Helm2DProblem, '_deviceItems')
for what, fn in (('dpred', lambda: sv.dpred(ctrue.ravel())), ('Jtvec', lambda: p.Jtvec(m, resid))):
    del LOG[:]; T00[0] = time.perf_counter()
    _zl.runtime_stats(reset=True)
    import torch
    tm0 = torch.cuda.memory_stats().get('num_device_alloc', 0), torch.cuda.memory_stats().get('num_device_free', 0)
    sys.stderr.write('[c4] %s starts\n' % what); sys.stderr.flush()
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable(); fn(); pr.disable(); dt = time.perf_counter() - t0
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18)
    sys.stderr.write('[c4] %s ends\n' % what); sys.stderr.flush()
    print('=== %s: %.3f s' % (what, dt)); print('   libhelm runtime objects:', _zl.runtime_stats()); print('   torch device allocs / frees:', torch.cuda.memory_stats().get('num_device_alloc', 0) - tm0[0], torch.cuda.memory_stats().get('num_device_free', 0) - tm0[1]); print('\n'.join(s.getvalue().splitlines()[:40]))
    for k, (c, t) in sorted(ACC.items(), key=lambda kv: -kv[1][1]): print('   %-46s %4d calls %8.1f ms' % (k, c, 1e3 * t))
    ACC.clear()
    print('   timeline (start ms, duration ms, thread, call) of the calls that took more than 0.3 ms:')
    for t_, d_, th_, nm_ in sorted(LOG):
        if d_ > 0.3: print('      %8.2f %8.2f  %-14s %s' % (t_, d_, th_, nm_))
