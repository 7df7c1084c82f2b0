import os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['HELM_ND_DEBUG'] = '3'
import torch
import zephyr_amd as za
from determinism_probe import device_solve
nz, nx, nrhs = 150, 170, 9
rng = np.random.default_rng(11)
c = 2500. + 500. * np.sin(np.arange(nz)[:, None] / 20.) * np.ones((nz, nx))
cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, freq=8., nPML=8, rtol=1e-10, method='direct', batch=256)
locs = np.stack([rng.uniform(100., 10. * nx - 100., nrhs), rng.uniform(20., 60., nrhs)], axis=1)
q = za.SparseKaiserSource(cfg)(locs).toarray()
for rep in range(6):
    op = za.MiniZephyr(cfg)
    h = hashlib.sha1(device_solve(op, q).tobytes()).hexdigest()[:6]
    sys.stderr.write('=== rep %d -> %s\n' % (rep, h)); sys.stderr.flush()
    del op.factors
