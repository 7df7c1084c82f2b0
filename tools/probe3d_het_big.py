#!/usr/bin/env python3
"""Development probe: the config-5 grid (256 x 256 x 128) with a strongly heterogeneous model (1500-5500 m/s), 8 sources at 3 and 5 Hz.
At 5 Hz the slowest cells have 30 points per wavelength: one coarsening would leave a 138 x 138 x 74 level to solve directly (115 GB of
plane inverses, over the budget), so the hierarchy goes one level deeper (7.5 points).  DESIGN.md section 5.3."""
import numpy as np, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zephyr_amd as za
nz, ny, nx, dx = 128, 256, 256, 10.
rng = np.random.default_rng(5)
iz, iy, ix = np.mgrid[0:nz, 0:ny, 0:nx]
c = 1500. + 20. * iz + 300. * np.sin(2 * np.pi * ix / 90.) * np.cos(2 * np.pi * iy / 70.) + 100. * rng.standard_normal((nz, ny, nx))
c[(iz > 60) & (iz < 75)] += 1200.
c = np.clip(c, 1500., 5500.)
rho = 1000. + 0.3 * (c - 1500.)
del iz, iy, ix
N = nz * ny * nx
q = np.zeros((8, N), complex).T
for s in range(8):
    q[((20 + 10 * s) * ny + 60 + 15 * s) * nx + 50 + 20 * s, s] = 1.
for f in (3., 5.):
    op = za.Helm3D(dict(nx=nx, ny=ny, nz=nz, dx=dx, c=c, rho=rho, freq=f, nPML=10, rtol=1e-8, maxit=3000, method='auto', batch=8))
    t0 = time.time(); u = op * q; dt = time.time() - t0
    r = op.applyForward(u.conj()) - q
    print('f=%.0f Hz (min ppw %.0f): %.2f s, iterations %s, true relres %.1e' % (f, c.min() / (f * dx), dt, [i['iterations'] for i in op.lastInfo], np.linalg.norm(r, axis=0).max()), flush=True)
    del op, u, r
