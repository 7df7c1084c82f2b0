#!/usr/bin/env python3
"""config 4's dpred(m) / Jtvec(m, v) called back to back, alternating two models: seconds of every call (a GPU stall shows as + 50-80 ms).  tools/c4_repeat.py [calls]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zephyr_amd as za
from zephyr_amd import _lib
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
models = [ccur.ravel(), ctrue.ravel()]
resid = sv.dpred(models[0]) - dobs
p.Jtvec(models[0], resid)
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 12
mode = sys.argv[2] if len(sys.argv) > 2 else 'alternate'
import torch
import threading, functools
LOG = []; LK = threading.Lock(); T0 = [0.0]
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    @functools.wraps(f)
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            dt = 1e3 * (time.perf_counter() - t0)
            if dt > 8.0:
                with LK: LOG.append('%8.1f ms +%6.1f ms %-12s %s' % (1e3 * (t0 - T0[0]), dt, threading.current_thread().name[:12], label or name))
    setattr(obj, name, w)
if os.environ.get('C4_TIMELINE'):
    from zephyr_amd import discretization
    for nm in ('prefactor', 'rhsFromSparseDevice', 'solveDevice', 'sampleDevice', 'imagingAccumulateDevice'):
        wrap(discretization.BaseDiscretization, nm)
    wrap(torch.Tensor, 'cpu', 'Tensor.cpu'); wrap(_lib, 'wait_torch_stream'); wrap(_lib, 'to_device'); wrap(torch, 'empty', 'torch.empty'); wrap(torch.cuda, 'synchronize', 'torch.cuda.synchronize')
    L = _lib.load()
    for nm in ('helm_solve_device', 'helm_sample_device', 'helm_rhs_from_coo_device_layout', 'helm_prefactor_n', 'helm_set_model', 'helm_assemble'):
        wrap(L, nm, 'C ' + nm)
T0[0] = time.perf_counter()
if mode.startswith('direct'):
    # the problem's own resident operators, driven like tools/stall_probe2.py does (no survey / problem code in the loop)
    from zephyr_amd import dispatch
    subs = p.system.subProblems
    N = n * n
    dev = torch.device('cuda', 0)
    R = torch.zeros((ns, N), dtype=torch.complex128, device=dev); U = torch.empty_like(R)
    R[:, N // 2] = 1.0
    torch.cuda.synchronize()
    ts = []
    for k in range(calls):
        t0 = time.perf_counter()
        if 'threads' in mode:
            items = [dispatch.WorkItem((lambda _p, op=op: op.solveDevice(R.data_ptr(), U.data_ptr(), ns, N)), None) for op in subs]
            pipes = dispatch.dispatch([(0, items)], lookahead=1)
            for it in items: it.future.result()
            for p_ in pipes: p_.join()
        else:
            for op in subs: op.solveDevice(R.data_ptr(), U.data_ptr(), ns, N)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    print('%-16s ms per call: %s' % (mode, ' '.join('%5.0f' % t for t in ts)))
    sys.exit(0)
for what in ('dpred', 'Jtvec'):
    ts = []
    t_begin = time.perf_counter()
    for k in range(calls):
        m = models[k % 2] if mode == 'alternate' else models[0]
        _lib.runtime_stats(reset=True)
        t0 = time.perf_counter()
        if what == 'dpred': sv.dpred(m)
        else: p.Jtvec(m, resid)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    st = _lib.runtime_stats()
    for l in LOG: print('   ', l)
    del LOG[:]
    print('%s %-9s ms per call: %s   total %.0f ms (last call: sync calls %d, slow %d, worst %.1f ms)' % (what, mode, ' '.join('%5.0f' % t for t in ts), 1e3 * (time.perf_counter() - t_begin), st['sync_calls'], st['slow_syncs'], st['worst_sync_ms']), flush=True)
