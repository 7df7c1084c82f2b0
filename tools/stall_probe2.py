#!/usr/bin/env python3
"""Stall hunt, part 2: eight RESIDENT operators (factors kept), a job = eight solves through the dispatcher like config 4's dpred on an unchanged model (19 ms);
ingredients of problem.py's work item switched on one at a time.   tools/stall_probe2.py <reps> <modes...>"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import scipy.sparse as sp
import zephyr_amd as za
from zephyr_amd import dispatch, _lib
from zephyr_amd.models import marmousi_like
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n, dx, nf, ns, nr = 512, 10.0, 8, 64, 128
N = n * n
c = marmousi_like(n, n, dx)
cfg = dict(nx=n, nz=n, dx=dx, dz=dx, c=c, nPML=10)
freqs = np.linspace(3.0, 10.0, nf)
locs = np.stack([np.linspace(200.0, 4920.0, ns), np.full(ns, 20.0)], axis=1)
rec = np.stack([np.linspace(100.0, dx * n - 100.0, nr), np.full(nr, 20.0)], axis=1)
qs = za.SparseKaiserSource(cfg)(locs)
Rm = sp.csr_matrix(za.SparseKaiserSource(cfg)(rec).T); Rm.sum_duplicates()
dev = torch.device('cuda', 0)
SETUP = os.environ.get('PROBE_SETUP', '')
if 'churn' in SETUP:                    # a process that has been through other models: sixteen operators prefactored, solved and destroyed first
    for f in list(freqs) * 2:
        o = za.Eurus(dict(cfg, freq=float(f) + 0.1, rtol=1e-10, batch=ns, device=0)); o.prefactor()
        if 'churnkeep' in SETUP: SETUP_KEEP = globals().setdefault('SETUP_KEEP', []); SETUP_KEEP.append(o)
        else: del o.factors
    for o in globals().get('SETUP_KEEP', []): del o.factors
ops = [za.Eurus(dict(cfg, freq=float(f), rtol=1e-10, batch=ns, device=0)) for f in freqs]
if 'pf' in SETUP:
    for op in ops: op.prefactor()
d_rhs = torch.from_numpy(np.ascontiguousarray(qs.toarray().T)).to(dev)
d_u = torch.empty((ns, N), dtype=torch.complex128, device=dev)
d_out = torch.empty((nr, ns), dtype=torch.complex128, device=dev)
csr = (_lib.to_device(Rm.indptr, dev, np.int64), _lib.to_device(Rm.indices, dev, np.int64), _lib.to_device(Rm.data, dev, np.complex128), nr)
for op in ops:
    op.solveDevice(d_rhs.data_ptr(), d_u.data_ptr(), ns, N)


def job(mode):
    state = {}

    def solve(_p, op=None):
        R, U, out = d_rhs, d_u, d_out
        if 'alloc' in mode and 'R' not in state:
            state['R'] = torch.empty((ns, N), dtype=torch.complex128, device=dev); state['U'] = torch.empty((ns, N), dtype=torch.complex128, device=dev)
            state['out'] = torch.empty((nr, ns), dtype=torch.complex128, device=dev)
        if 'alloc' in mode:
            R, U, out = state['R'], state['U'], state['out']
        if 'fromcoo' in mode or 'alloc' in mode:
            op.rhsFromSparseDevice(qs, R.data_ptr())
        op.solveDevice(R.data_ptr(), U.data_ptr(), ns, N)
        if 'sample' in mode:
            op.sampleDevice(U.data_ptr(), ns, csr, out.data_ptr())
        if 'd2h' in mode:
            x = out.cpu().numpy()
        return 0
    if 'threads' in mode:                       # the dispatcher's two threads, started per job like problem.py does
        items = [dispatch.WorkItem((lambda _p, op=op: solve(_p, op)), None) for op in ops]
        pipes = dispatch.dispatch([(0, items)], lookahead=1)
        for it in items:
            it.future.result()
        for p_ in pipes:
            p_.join()
    else:
        for op in ops:
            solve(None, op)


for mode in sys.argv[2:] or ['plain', 'threads', 'threads+sample+d2h', 'threads+alloc+sample+d2h']:
    job(mode); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        job(mode)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    print('%-28s ms per job: %s' % (mode, ' '.join('%5.0f' % t for t in ts)), flush=True)
