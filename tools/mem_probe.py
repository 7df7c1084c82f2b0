#!/usr/bin/env python3
"""Device memory after every pass of the 16-item bench job through the device pipeline (free GB by hipMemGetInfo): does the pool keep growing?  tools/mem_probe.py [group] [passes]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zephyr_amd as za
from zephyr_amd import dispatch, _lib
from zephyr_amd.models import marmousi_like
group = int(sys.argv[1]) if len(sys.argv) > 1 else 2
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n, dx, ns = 1024, 9.0, 256
N = n * n
c = marmousi_like(n, n, dx).astype(np.complex128)
cfg = dict(nx=n, nz=n, dx=dx, dz=dx, c=c, nPML=10)
freqs = np.linspace(2.0, 9.5, 16)
locs = np.stack([np.linspace(0.04 * n * dx, 0.96 * n * dx, ns), np.full(ns, 20.0)], axis=1)
q = za.SparseKaiserSource(cfg)(locs).toarray()
dev = torch.device('cuda', 0)
d_rhs = torch.from_numpy(np.ascontiguousarray(q)).to(dev)
d_u = torch.empty((N, ns), dtype=torch.complex128, device=dev)


def job():
    def prep(f):
        op = za.Eurus(dict(cfg, freq=float(f), rtol=1e-10, batch=ns, device=0))
        op.handle
        if group <= 1: op.prefactor()
        return op

    def solve(op):
        op.solveDevice(d_rhs.data_ptr(), d_u.data_ptr(), ns, N, layout='node')
        del op.factors
        return 0
    items = [dispatch.WorkItem(solve, (lambda f=f: prep(f))) for f in freqs]
    return list(dispatch.pipelined(items, device=0, lookahead=1, group=group, group_prepare=za.prefactor_many if group > 1 else None))


for p in range(passes):
    _lib.runtime_stats(reset=True)
    t0 = time.perf_counter(); job(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    free, tot = torch.cuda.mem_get_info()
    st = _lib.runtime_stats()
    print('pass %d: %.3f s; free %.1f of %.1f GB; device allocs in this pass %d (%.1f GB), frees %d' % (p, dt, free / 1e9, tot / 1e9, st['dev_allocs'], st['dev_alloc_bytes'] / 1e9, st['dev_frees']), flush=True)
