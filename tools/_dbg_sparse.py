import os, sys, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import zephyr_amd as za
nsrc = int(sys.argv[1]) if len(sys.argv) > 1 else 70
nz, nx = 150, 170
rng = np.random.default_rng(nsrc)
c = 1800. + 2000. * rng.random((nz, nx))
cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, freq=8., nPML=8, rtol=1e-10, method='direct', batch=256)
locs = np.stack([rng.uniform(100., 10. * nx - 100., nsrc), rng.uniform(20., 60., nsrc)], axis=1)
q = za.SparseKaiserSource(cfg)(locs).toarray()
q[:, 1] = 0.0
if nsrc > 64:
    q[:, 64:128] = 0.0
    q[:, -1] = rng.standard_normal(nz * nx) + 1j * rng.standard_normal(nz * nx)
out = {}
for mode in ('1', '0'):
    os.environ['HELM_ND_SPARSE_RHS'] = mode
    op = za.Eurus(cfg)
    out[mode] = op * q
    print(mode, [ (i['iterations'], '%.1e' % i['relres']) for i in op.lastInfo][:3], [ (i['iterations'], '%.1e' % i['relres']) for i in op.lastInfo][-3:])
    del op.factors
d = np.abs(out['1'] - out['0'])
cols = np.where(d.max(axis=0) > 0)[0]
print('differing columns', cols[:20], len(cols), 'max diff', d.max(), 'scale', np.abs(out['0']).max())
if len(cols):
    j = cols[0]; cells = np.where(d[:, j] > 0)[0]
    print('col', j, 'cells differing', len(cells), [(int(x) // nx, int(x) % nx) for x in cells[:10]])
