"""config-4 operators one by one (GPU box): factorisation / solve milliseconds, passes and treated fronts per frequency on the smoothed 512^2 model."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zephyr_amd as za
from zephyr_amd.models import marmousi_like, box_smooth
n, dx, ns = 512, 10.0, 64
ctrue = marmousi_like(n, n, dx)
for name, c in (('true', ctrue), ('smoothed', box_smooth(ctrue, 12))):
    for f in np.linspace(3.0, 10.0, 8):
        cfg = dict(nx=n, nz=n, dx=dx, dz=dx, c=c, freq=float(f), rtol=1e-10, method='direct', batch=ns)
        src = np.stack([np.linspace(200.0, 4920.0, ns), np.full(ns, 20.0)], axis=1)
        q = np.ascontiguousarray(za.SparseKaiserSource(cfg)(src).toarray())
        R = torch.from_numpy(q).cuda(); U = torch.empty_like(R)
        op = za.Eurus(cfg); op.setProfiling(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        op.solveDevice(R.data_ptr(), U.data_ptr(), ns, n * n, layout='node')
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        t = op.lastTiming(); info = op.lastInfo
        print('%-8s f %5.2f  wall %6.2f ms  factor %6.2f  solve_call %6.2f  passes %d  relres %.1e' % (name, f, 1e3 * dt, t['factor_ms'], t['solve_ms'], max(i['iterations'] for i in info), max(i['relres'] for i in info)), flush=True)
        del op.factors
