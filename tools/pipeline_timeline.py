#!/usr/bin/env python3
"""Round 6: host timeline of the headline's device pipeline (1024^2 Eurus, 256 sources per item; sets of two, lookahead as bench.py): when each item's build /
handle creation (helm_create, helm_set_model, helm_assemble) / helm_prefactor_many / helm_solve_device starts and ends on the two threads.
   python3 tools/pipeline_timeline.py [items] [lookahead]"""
import os, sys, time, threading
os.environ.setdefault('OPENBLAS_NUM_THREADS', '1'); os.environ.setdefault('OMP_NUM_THREADS', '1'); os.environ.setdefault('HELM_POOL_SPARE_AUTO', '0')
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from zephyr_amd import Eurus, SparseKaiserSource, dispatch, prefactor_many, _lib
n, dx, B = 1024, 9.0, 256
nitems = int(sys.argv[1]) if len(sys.argv) > 1 else 14
la = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = bench.build_config(n, dx)
freqs = np.linspace(2.0, 9.5, 16)
locs = np.stack([np.linspace(300.0, dx * n - 300.0, B), np.full(B, 20.0)], axis=1)
q = SparseKaiserSource(cfg)(locs)
dev = torch.device('cuda', 0)
N = n * n
d_rhs = _lib.to_device(np.ascontiguousarray(q.toarray()), dev)
d_u = torch.empty((N, B), dtype=torch.complex128, device=dev)
LOG, LK, T0 = [], threading.Lock(), [0.0]
def mark(what, t0, t1):
    with LK: LOG.append((1e3 * (t0 - T0[0]), 1e3 * (t1 - T0[0]), threading.current_thread().name[-11:], what))
_L = _lib.load()
class _Timed(object):
    def __init__(self, name): self.f = getattr(_L, name); self.name = name
    def __call__(self, *a):
        t0 = time.perf_counter()
        try: return self.f(*a)
        finally:
            t1 = time.perf_counter()
            if t1 - t0 > 2e-4: mark('  C ' + self.name, t0, t1)
for nm in ('helm_create', 'helm_set_model', 'helm_assemble', 'helm_destroy', 'helm_prefactor_many', 'helm_solve_device'):
    setattr(_L, nm, _Timed(nm))
def prep(w):
    t0 = time.perf_counter()
    op = Eurus(dict(cfg, freq=float(freqs[w % 16]), rtol=1e-10, maxit=400000, method='auto', batch=B, device=0))
    op.setProfiling(True)
    mark('build %d' % w, t0, time.perf_counter())
    return op
def solve(w, op):
    t0 = time.perf_counter()
    op.solveDevice(d_rhs.data_ptr(), d_u.data_ptr(), B, N, layout='node')
    del op.factors
    mark('solve %d' % w, t0, time.perf_counter())
def run(ws):
    items = [dispatch.WorkItem((lambda op, w=w: solve(w, op)), (lambda w=w: prep(w))) for w in ws]
    return list(dispatch.pipelined(items, device=0, lookahead=la, group=2, group_prepare=prefactor_many))
run(range(5)); _L.helm_pool_spares(0, 3); torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
for rep in range(2):
    del LOG[:]; torch.cuda.synchronize(); T0[0] = t0 = time.perf_counter()
    run(range(5, 5 + nitems)); torch.cuda.synchronize()
    print('--- %d items in %.1f ms (%.0f wavefields/s), lookahead %d' % (nitems, 1e3 * (time.perf_counter() - t0), nitems * B / (time.perf_counter() - t0), la))
    for a, b, th, what in sorted(LOG):
        print('   %7.2f .. %7.2f  (%6.2f)  %-11s %s' % (a, b, b - a, th, what))
    _L.helm_pool_spares(0, 3)
