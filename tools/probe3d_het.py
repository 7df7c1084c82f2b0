#!/usr/bin/env python3
"""Development probe: 3-D solve on a strongly heterogeneous 128 x 128 x 64 model (1500-5000 m/s: gradient, lateral undulation, a fast
layer, noise), layer-preserving hierarchy (HELM_MG3_KEEP=1) against the standard shifted cycle (0).  DESIGN.md section 5.3."""
import numpy as np, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zephyr_amd as za
nz, ny, nx, dx = 64, 128, 128, 10.
rng = np.random.default_rng(5)
iz, iy, ix = np.mgrid[0:nz, 0:ny, 0:nx]
c = 1500. + 35. * iz + 300. * np.sin(2 * np.pi * ix / 60.) * np.cos(2 * np.pi * iy / 45.) + 150. * rng.standard_normal((nz, ny, nx))
c[(iz > 30) & (iz < 40)] += 1200.           # fast layer
c = np.clip(c, 1500., 5000.)
rho = 1000. + 0.3 * (c - 1500.)
N = nz * ny * nx
q = np.zeros((4, N), complex).T
for s in range(4):
    q[((15 + 8 * s) * ny + 40 + 15 * s) * nx + 30 + 20 * s, s] = 1.
for f in (3., 4., 6.):
    for keep in ('1', '0'):
        os.environ['HELM_MG3_KEEP'] = keep
        op = za.Helm3D(dict(nx=nx, ny=ny, nz=nz, dx=dx, c=c, rho=rho, freq=f, nPML=10, rtol=1e-8, maxit=20000, method='auto', batch=4))
        t0 = time.time(); u = op * q; dt = time.time() - t0
        r = op.applyForward(u.conj()) - q
        print('f=%.0f Hz (min ppw %.0f) keep=%s: %.2f s, iterations %s, true relres %.1e' % (f, c.min() / (f * dx), keep, dt, [i['iterations'] for i in op.lastInfo], np.linalg.norm(r, axis=0).max() / 1.0), flush=True)
        del op
