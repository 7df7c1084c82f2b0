#!/usr/bin/env python3
"""Milliseconds of the factorisation of 1, 2 and 4 operators of the 1024^2 bench model, one after the other against in the same launches (helm_prefactor_many),
with nothing else on the GPU.   tools/factor_many_probe.py [n = 1024] [reps = 5]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zephyr_amd as za
from zephyr_amd.models import marmousi_like
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dx = 9.0 if n == 1024 else 10.0
c = marmousi_like(n, n, dx).astype(np.complex128)
cfg = dict(nx=n, nz=n, dx=dx, dz=dx, c=c, nPML=10, rtol=1e-10, method='direct', batch=256)
freqs = [3.5, 5.5, 7.5, 9.5]


def run(nf, together):
    ts = []
    for r in range(reps + 1):
        ops = [za.Eurus(dict(cfg, freq=f + 0.01 * r)) for f in freqs[:nf]]
        for op in ops: op.handle            # create + assemble
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if together: za.prefactor_many(ops)
        else:
            for op in ops: op.prefactor()
        torch.cuda.synchronize()
        if r: ts.append(1e3 * (time.perf_counter() - t0))
        for op in ops: del op.factors
    return np.median(ts), min(ts)


for nf in (1, 2, 4):
    a = run(nf, False)
    b = run(nf, True) if nf > 1 else a
    print('%d operator(s): one after the other %.2f ms (min %.2f) = %.2f per operator; in the same launches %.2f ms (min %.2f) = %.2f per operator' % (nf, a[0], a[1], a[0] / nf, b[0], b[1], b[0] / nf), flush=True)
