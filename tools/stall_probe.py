#!/usr/bin/env python3
"""Which call of the config-4 work items makes the GPU sit idle for 50-80 ms once per job?  The config-2 job (8 operators at 512^2 through the device pipeline)
with one ingredient of config 4's items added at a time; wall time of every repetition (a stall shows as + 50 ms).   tools/stall_probe.py [reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zephyr_amd as za
from zephyr_amd import dispatch, _lib
from zephyr_amd.models import marmousi_like
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n, dx, nf, ns = 512, 10.0, 8, 64
N = n * n
c = marmousi_like(n, n, dx)
cfg = dict(nx=n, nz=n, dx=dx, dz=dx, c=c, nPML=10)
freqs = np.linspace(3.0, 10.0, nf)
locs = np.stack([np.linspace(200.0, 4920.0, ns), np.full(ns, 20.0)], axis=1)
qs = za.SparseKaiserSource(cfg)(locs)
dev = torch.device('cuda', 0)
d_rhs = torch.from_numpy(np.ascontiguousarray(qs.toarray().T)).to(dev)
d_u = torch.empty((ns, N), dtype=torch.complex128, device=dev)
big = np.random.default_rng(0).standard_normal(N * 2).view(np.complex128)          # 4 MB of pageable memory
small = np.arange(1000, dtype=np.int64)


KEPT = []


def job(mode):
    if 'keep' in mode:                     # config 4 keeps a model's eight operators (factors resident) until the next model arrives, then drops them all at once
        for o in KEPT:
            del o.factors
        del KEPT[:]

    def prep(f):
        op = za.Eurus(dict(cfg, freq=float(f), rtol=1e-10, batch=ns, device=0))
        op.prefactor()
        return op

    def solve(op):
        if 'h2d_small' in mode:
            t = torch.from_numpy(small).to(dev)
        if 'h2d_big' in mode:
            t = torch.from_numpy(big.copy()).to(dev)
        if 'h2d_big_staged' in mode:
            t = _lib.to_device(big.copy(), dev)
        if 'tsync' in mode:
            torch.cuda.current_stream(dev).synchronize()
        if 'empty' in mode:
            e = torch.empty((ns, N), dtype=torch.complex128, device=dev)
        if 'fromcoo' in mode:
            op.rhsFromSparseDevice(qs, d_rhs.data_ptr())
        op.solveDevice(d_rhs.data_ptr(), d_u.data_ptr(), ns, N)
        if 'd2h' in mode:
            x = d_u[:2, :128].cpu().numpy()
        if 'newmodel' in mode:
            pass
        if 'keep' in mode:
            KEPT.append(op)
        else:
            del op.factors
        return 0
    items = [dispatch.WorkItem(solve, (lambda f=f: prep(f))) for f in freqs]
    return list(dispatch.pipelined(items, device=0, lookahead=1))


IDLE = float(os.environ.get('PROBE_IDLE_MS', '0')) * 1e-3       # host-only pause before every job: does a GPU that has been idle stall when work comes back?
for mode in (sys.argv[2:] or ['plain', 'tsync', 'h2d_small', 'h2d_big', 'h2d_big_staged', 'empty', 'fromcoo', 'd2h', 'fromcoo+d2h+tsync']):
    job(mode)
    torch.cuda.synchronize()
    ts = []
    import ctypes
    _lib.load().helm_debug_stall_watch(1, None, None)
    for _ in range(reps):
        _lib.runtime_stats(reset=True)
        if IDLE > 0: time.sleep(IDLE)
        t0 = time.perf_counter()
        job(mode)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    st = _lib.runtime_stats()
    wg, wn = ctypes.c_double(0), ctypes.c_longlong(0)
    _lib.load().helm_debug_stall_watch(0, ctypes.byref(wg), ctypes.byref(wn))
    print('   clock-reading thread: worst gap %.2f ms, %d gaps over 5 ms' % (wg.value, wn.value))
    print('%-22s wall ms per job: %s   (last job: slow syncs %d, worst %.1f ms)' % (mode, ' '.join('%6.1f' % t for t in ts), st['slow_syncs'], st['worst_sync_ms']), flush=True)
