#!/usr/bin/env python3
"""Round 6: host timeline of config 4's dpred(m) calls (512^2, 8 frequencies x 64 sources) -- when each item's prepare / set-prefactor / solve step starts and ends
on the dispatcher's threads, call after call on alternating models, to see what a slow call does differently.   python3 tools/c4_timeline.py [calls]"""
import os, sys, time, threading
os.environ.setdefault('OPENBLAS_NUM_THREADS', '1'); os.environ.setdefault('OMP_NUM_THREADS', '1')
def _numa_report():
    import glob
    nodes = sorted(glob.glob('/sys/devices/system/node/node[0-9]*'))
    try: nb = open('/proc/sys/kernel/numa_balancing').read().strip()
    except Exception: nb = '?'
    print('numa nodes %d, numa_balancing %s, affinity %d cpus, on cpu %s' % (len(nodes), nb, len(os.sched_getaffinity(0)), open('/proc/self/stat').read().split()[38]))
    return nodes
def _parse(cl):
    out = set()
    for part in cl.strip().split(','):
        if '-' in part: a, b = part.split('-'); out.update(range(int(a), int(b) + 1))
        elif part: out.add(int(part))
    return out
_nodes = _numa_report()
if os.environ.get('C4_PIN') == '1' and _nodes:
    cur = int(open('/proc/self/stat').read().split()[38])
    for nd in _nodes:
        cpus = _parse(open(nd + '/cpulist').read())
        if cur in cpus:
            os.sched_setaffinity(0, cpus & os.sched_getaffinity(0)); print('pinned to', nd, len(cpus), 'cpus'); break
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zephyr_amd as za
from zephyr_amd import dispatch, discretization, _lib
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
mcur, mtrue = ccur.ravel(), ctrue.ravel()
d = sv.dpred(mcur); resid = d - dobs
g = p.Jtvec(mcur, resid)
LOG, LK, T0 = [], threading.Lock(), [0.0]
def mark(what, t0, t1):
    with LK: LOG.append((1e3 * (t0 - T0[0]), 1e3 * (t1 - T0[0]), threading.current_thread().name[-12:], what))
_init = dispatch.WorkItem.__init__
def init(self, solve, prepare=None):
    k = len([0 for _ in ITEMS]); ITEMS.append(self)
    def s(x, solve=solve, k=k):
        t0 = time.perf_counter()
        try: return solve(x)
        finally: mark('solve %d' % k, t0, time.perf_counter())
    def pr(prepare=prepare, k=k):
        t0 = time.perf_counter()
        try: return prepare()
        finally: mark('prepare %d' % k, t0, time.perf_counter())
    _init(self, s, pr if prepare is not None else None)
dispatch.WorkItem.__init__ = init
ITEMS = []
_pm = discretization.prefactor_many
def pm(ops):
    t0 = time.perf_counter()
    try: return _pm(ops)
    finally: mark('prefactor_many x%d' % len(ops), t0, time.perf_counter())
discretization.prefactor_many = pm
_L = _lib.load()
class _Timed(object):
    'wall-clock marks around a ctypes entry point (the attribute on the CDLL object is replaced by this callable)'
    def __init__(self, name):
        self.f = getattr(_L, name); self.name = name
    def __call__(self, *a):
        t0 = time.perf_counter()
        try: return self.f(*a)
        finally:
            t1 = time.perf_counter()
            if t1 - t0 > 1e-3: mark('   C: %s' % self.name, t0, t1)
for nm in ('helm_create', 'helm_set_model', 'helm_assemble', 'helm_destroy', 'helm_prefactor_many', 'helm_solve_device', 'helm_rhs_from_coo_device', 'helm_sample_device'):
    setattr(_L, nm, _Timed(nm))
_ma = za.Eurus._model_arrays
def ma(self):
    t0 = time.perf_counter()
    try: return _ma(self)
    finally: mark('   py: _model_arrays', t0, time.perf_counter())
za.Eurus._model_arrays = ma
import zephyr_amd.problem as zp
if os.environ.get('C4_HEARTBEAT'):
    # (experiment) a thread that keeps the GPU from going idle: one tiny kernel every C4_HEARTBEAT milliseconds on a stream of its own
    _hb_stop = [False]
    def _hb():
        st = torch.cuda.Stream()
        x = torch.zeros(64, device='cuda')
        per = float(os.environ['C4_HEARTBEAT']) * 1e-3
        while not _hb_stop[0]:
            with torch.cuda.stream(st):
                x.add_(1.0)
            time.sleep(per)
    threading.Thread(target=_hb, daemon=True, name='heartbeat').start()
ncalls = int(sys.argv[1]) if len(sys.argv) > 1 else 6
import gc; gc.collect(); gc.freeze()
for c in range(ncalls):
    p.updateModel(mtrue)
    torch.cuda.synchronize(); del LOG[:]; del ITEMS[:]; _lib.runtime_stats(reset=True)
    T0[0] = t0 = time.perf_counter()
    sv.dpred(mcur)
    torch.cuda.synchronize(); t1 = time.perf_counter(); rs = _lib.runtime_stats()
    print('    runtime objects created in the call: streams %d, events %d, device allocations %d, pinned %d' % (rs['streams_created'], rs['events_created'], rs['dev_allocs'], rs['host_allocs']))
    print('--- dpred call %d: %.1f ms   (starts at CLOCK_MONOTONIC %d ns)' % (c, 1e3 * (t1 - t0), int(t0 * 1e9)))
    for a, b, th, what in sorted(LOG):
        print('   %7.2f .. %7.2f  (%6.2f)  %-12s %s' % (a, b, b - a, th, what))
    p.updateModel(mtrue)
    torch.cuda.synchronize()
    g = p.Jtvec(mcur, resid)
    torch.cuda.synchronize()
