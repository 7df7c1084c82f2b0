#!/usr/bin/env python3
"""Development probe: iteration counts / timings of the GPU solver on the bench model for chosen sources."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.build()
from zephyr_amd import Eurus, MiniZephyr, SparseKaiserSource, SimpleSource
from zephyr_amd.models import marmousi_like

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=1024); ap.add_argument('--dx', type=float, default=9.0)
ap.add_argument('--freqs', type=float, nargs='+', default=[4.5])
ap.add_argument('--method', default='mg'); ap.add_argument('--kind', default='eurus')
ap.add_argument('--src', default='point')   # point | kaiser
ap.add_argument('--nsrc', type=int, default=4); ap.add_argument('--zsrc', type=float, default=None)
ap.add_argument('--maxit', type=int, default=3000); ap.add_argument('--batch', type=int, default=8)
a = ap.parse_args()
n, dx = a.n, a.dx
c = marmousi_like(n, n, dx)
for f in a.freqs:
    cfg = dict(nx=n, nz=n, dx=dx, dz=dx, c=c, freq=f, rtol=1e-10, maxit=a.maxit, method=a.method, batch=a.batch)
    zs = a.zsrc if a.zsrc is not None else (12 * dx if a.src == 'point' else 20.0)
    xs = np.linspace(0.04 * n * dx, 0.96 * n * dx, a.nsrc) if a.nsrc > 1 else np.array([n // 2 * dx])
    locs = np.stack([xs, np.full(len(xs), zs)], 1)
    q = (SimpleSource(cfg)(locs) if a.src == 'point' else SparseKaiserSource(cfg)(locs).toarray())
    op = (Eurus if a.kind == 'eurus' else MiniZephyr)(cfg)
    t0 = time.time()
    try:
        u = op * q
        ok = 'ok'
    except ArithmeticError as e:
        ok = 'FAILED: %s' % e
    dt = time.time() - t0
    print('n', n, 'f', f, a.kind, a.method, a.src, 'z=%g' % zs, 'time %.2fs' % dt, ok)
    for loc, i in zip(locs, op.lastInfo):
        print('    x=%7.1f  its %5d restarts %d status %d relres %.2e' % (loc[0], i['iterations'], i['restarts'], i['status'], i['relres']))
