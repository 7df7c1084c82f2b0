"""Per-level device times (HELM_ND_TRACE=1) of one frequency of the headline workload: factorisation, forward and backward sweeps, warm."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zephyr_amd as za
from zephyr_amd.models import marmousi_like
n, dx, nsrc = 1024, 9.0, 256
freq = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
c = marmousi_like(n, n, dx)
cfg = dict(nx=n, nz=n, dx=dx, dz=dx, c=c, freq=freq, nPML=10, rtol=1e-10, method='direct', batch=256)
locs = np.stack([np.linspace(300., dx * n - 300., nsrc), np.full(nsrc, 20.)], axis=1)
q = np.ascontiguousarray(za.SparseKaiserSource(cfg)(locs).toarray())
R = torch.from_numpy(q).cuda()
U = torch.empty_like(R)
for rep in range(3):
    if rep == 2:
        os.environ['HELM_ND_TRACE'] = '1'
    op = za.Eurus(cfg)
    op.solveDevice(R.data_ptr(), U.data_ptr(), nsrc, n * n, layout='node')
    torch.cuda.synchronize()
    del op.factors
