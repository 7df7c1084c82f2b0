#!/usr/bin/env python3
"""Config 5 probe: 3-D 27-point operator, 256 x 256 x 128, homogeneous c = 2000 m/s, h = 10 m.
Times the batched apply (algorithmic bytes N*(32*B + 432)) and Jacobi-BiCGSTAB solves."""
import argparse, ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build()
from zephyr_amd import Helm3D, _lib

ap = argparse.ArgumentParser()
ap.add_argument('--nx', type=int, default=256); ap.add_argument('--ny', type=int, default=256); ap.add_argument('--nz', type=int, default=128)
ap.add_argument('--freqs', type=float, nargs='+', default=[2., 5.]); ap.add_argument('--nsrc', type=int, default=4)
ap.add_argument('--maxit', type=int, default=60000); ap.add_argument('--rtol', type=float, default=1e-8)
ap.add_argument('--no-solve', action='store_true'); ap.add_argument('--standard-cycle', action='store_true', help='HELM_MG3_KEEP=0: the shifted cycle with standard coarsening (round-2 baseline)'); ap.add_argument('--method', default='auto'); ap.add_argument('--no-apply', action='store_true')
a = ap.parse_args()
nx, ny, nz = a.nx, a.ny, a.nz
if a.standard_cycle:
    os.environ['HELM_MG3_KEEP'] = '0'
N = nx * ny * nz
cfg = dict(nx=nx, ny=ny, nz=nz, dx=10., c=2000., rho=1., freq=a.freqs[0], nPML=10, rtol=a.rtol, maxit=a.maxit, batch=a.nsrc, method=a.method)
op = Helm3D(cfg)
lib = _lib.load()
dev = torch.device('cuda', 0)
out = {'grid': [nz, ny, nx], 'preconditioner': 'standard shifted cycle' if a.standard_cycle else 'layer-preserving hierarchy + direct coarse solve (column dissection / plane-by-plane elimination)', 'apply': []}
for B in (() if a.no_apply else (1, 4, 8, 16)):
    X = torch.randn((B, N), dtype=torch.complex128, device=dev)
    Y = torch.empty_like(X)
    torch.cuda.synchronize()
    op.setProfiling(True)
    ms, bytes_ = 0.0, 0.0
    for rep in range(6):
        _lib.check(lib.helm_apply_device(op.handle, 0, 0, ctypes.c_void_p(X.data_ptr()), ctypes.c_void_p(Y.data_ptr()), B), op.handle)
        t = op.lastTiming()
        if rep >= 1:
            ms += t['apply_ms']; bytes_ += t['apply_bytes']
    out['apply'].append({'B': B, 'us': 1e3 * ms / 5, 'GBps_algorithmic': bytes_ / (ms * 1e-3) / 1e9})
    print('apply B=%2d  %8.1f us  %7.1f GB/s (N*(32B+432))' % (B, 1e3 * ms / 5, bytes_ / (ms * 1e-3) / 1e9), flush=True)
    del X, Y
op.setProfiling(False)
if not a.no_solve:
    out['solve'] = []
    for f in a.freqs:
        cfg['freq'] = f
        op = Helm3D(cfg)
        # right-hand sides kept source-major in memory: the (N, nsrc) view the operator takes is then free of copies
        q = np.zeros((a.nsrc, N), complex).T
        for s in range(a.nsrc):
            q[((20 + 5 * s) * ny + ny // 2) * nx + nx // 4 + (30 * s) % (nx // 2), s] = 1.     # all inside the physical domain
        t0 = time.time()
        try:
            u = op * q
            status = 'ok'
        except ArithmeticError as e:
            status = str(e)
        dt = time.time() - t0
        its = [i['iterations'] for i in op.lastInfo]
        rec = {'freq': f, 'nsrc': a.nsrc, 'seconds': dt, 'iterations': its, 'status': status}
        del u
        # the same solves with right-hand sides and wavefields resident in HBM (preconditioner set-up included: a fresh operator)
        del op
        op = Helm3D(cfg)
        Q = torch.from_numpy(np.ascontiguousarray(q.T)).to(dev)
        U = torch.empty_like(Q)
        torch.cuda.synchronize()
        t0 = time.time()
        try:
            op.solveDevice(Q.data_ptr(), U.data_ptr(), a.nsrc)
        except ArithmeticError as e:
            rec['status_device'] = str(e)
        torch.cuda.synchronize()
        rec['seconds_device_resident'] = time.time() - t0
        t0 = time.time()
        op.solveDevice(Q.data_ptr(), U.data_ptr(), a.nsrc)         # preconditioner already built for this frequency
        torch.cuda.synchronize()
        rec['seconds_device_resident_reusing_setup'] = time.time() - t0
        del Q, U
        out['solve'].append(rec)
        print('solve f=%g Hz nsrc=%d  host buffers %.2f s | HBM-resident %.2f s (of which set-up %.2f s)  its %s  %s'
              % (f, a.nsrc, dt, rec['seconds_device_resident'], rec['seconds_device_resident'] - rec['seconds_device_resident_reusing_setup'], its, status), flush=True)
        del op
print(json.dumps(out))
