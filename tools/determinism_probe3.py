import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['HELM_ND_DEBUG'] = '2'
import torch
import zephyr_amd as za
from zephyr_amd.models import marmousi_like
from determinism_probe import device_solve
def run(cls, n, dx, freq, c):
    sys.stderr.write('##### %s n=%d f=%g\n' % (cls, n, freq)); sys.stderr.flush()
    cfg = dict(nx=n, nz=n, dx=dx, dz=dx, c=c, freq=freq, nPML=10, rtol=1e-10, method='direct', batch=256)
    locs = np.stack([np.linspace(300., dx * n - 300., 8), np.full(8, 20.)], axis=1)
    q = za.SparseKaiserSource(cfg)(locs).toarray()
    op = getattr(za, cls)(cfg)
    device_solve(op, q)
    sys.stderr.write('   relres %.2e passes %d\n' % (max(i['relres'] for i in op.lastInfo), max(i['iterations'] for i in op.lastInfo)))
    del op.factors
nz = 160
c_s = 2500. + 500. * np.sin(np.arange(nz)[:, None] / 20.) * np.ones((nz, nz))
run('MiniZephyr', nz, 10., 8., c_s)
run('Eurus', nz, 10., 8., c_s)
run('MiniZephyr', 512, 9., 8., marmousi_like(512, 512, 9.))
run('Eurus', 512, 9., 8., marmousi_like(512, 512, 9.))
