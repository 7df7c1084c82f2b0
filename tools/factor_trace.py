#!/usr/bin/env python3
"""Per-level device time of a warm factorisation at 1024^2: one operator, and two / four in the same launches (HELM_ND_TRACE=1).  tools/factor_trace.py"""
import os, sys
os.environ['HELM_ND_TRACE'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import zephyr_amd as za
from zephyr_amd.models import marmousi_like
n = 1024
c = marmousi_like(n, n, 9.0).astype(np.complex128)
cfg = dict(nx=n, nz=n, dx=9.0, dz=9.0, c=c, nPML=10, rtol=1e-10, method='direct', batch=256)
for rnd in range(2):            # the second round is the warm one
    for nf in (1, 2, 4):
        ops = [za.Eurus(dict(cfg, freq=f + 0.01 * rnd)) for f in [3.5, 5.5, 7.5, 9.5][:nf]]
        for op in ops: op.handle
        torch.cuda.synchronize()
        sys.stderr.write('=== round %d: %d operator(s)\n' % (rnd, nf)); sys.stderr.flush()
        za.prefactor_many(ops) if nf > 1 else ops[0].prefactor()
        torch.cuda.synchronize()
        for op in ops: del op.factors
