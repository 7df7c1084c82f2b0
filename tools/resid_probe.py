"""Residual launch of the node-major direct path at the headline size under a few switch settings (GPU box): apply_ms / launches from lastTiming()."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zephyr_amd as za
from zephyr_amd.models import marmousi_like
n, dx, nsrc = 1024, 9.0, 256
c = marmousi_like(n, n, dx)
cfg = dict(nx=n, nz=n, dx=dx, dz=dx, c=c, freq=5.0, nPML=10, rtol=1e-10, method='direct', batch=256)
locs = np.stack([np.linspace(300., dx * n - 300., nsrc), np.full(nsrc, 20.)], axis=1)
q = np.ascontiguousarray(za.SparseKaiserSource(cfg)(locs).toarray())
R = torch.from_numpy(q).cuda()
U = torch.empty_like(R)
for env in ({}, {'HELM_ND_DIRECT_OUT': '0'}, {'HELM_ND_SPARSE_RHS': '0'}, {'HELM_ND_SPARSE_RHS': '0', 'HELM_ND_DIRECT_OUT': '0'}):
    for k in ('HELM_ND_DIRECT_OUT', 'HELM_ND_SPARSE_RHS'):
        os.environ.pop(k, None)
    os.environ.update(env)
    op = za.Eurus(cfg)
    op.setProfiling(True)
    out = []
    for rep in range(3):
        op.solveDevice(R.data_ptr(), U.data_ptr(), nsrc, n * n, layout='node')
        torch.cuda.synchronize()
        t = op.lastTiming()
        out.append((round(t['apply_ms'], 3), int(t['apply_launches']), round(t['solve_ms'], 2)))
    print(env, out, flush=True)
    del op.factors
