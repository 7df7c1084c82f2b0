#!/usr/bin/env python3
"""27-point apply alone, config-5 grid (256 x 256 x 128), one batch width: the launch the PMC passes of tools/run_profiles_r5.sh count HBM bytes of.
usage: apply3d_micro.py [B = 16] [reps = 6]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build()
from zephyr_amd import Helm3D, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
nx, ny, nz = 256, 256, 128
N = nx * ny * nz
op = Helm3D(dict(nx=nx, ny=ny, nz=nz, dx=10., c=2000., rho=1., freq=5., nPML=10))
lib = _lib.load()
dev = torch.device('cuda', 0)
X = torch.randn((B, N), dtype=torch.complex128, device=dev)
Y = torch.empty_like(X)
torch.cuda.synchronize()
op.setProfiling(True)
ms = by = 0.0
for rep in range(reps):
    _lib.check(lib.helm_apply_device(op.handle, 0, 0, ctypes.c_void_p(X.data_ptr()), ctypes.c_void_p(Y.data_ptr()), B), op.handle)
    t = op.lastTiming()
    if rep:
        ms += t['apply_ms']; by += t['apply_bytes']
n = max(1, reps - 1)
print(json.dumps({'B': B, 'grid_nz_ny_nx': [nz, ny, nx], 'us': 1e3 * ms / n, 'algorithmic_bytes_per_launch': by / n, 'formula': 'N*(32*B + 432)',
                  'GBps': by / (ms * 1e-3) / 1e9, 'frac_of_8TBps': by / (ms * 1e-3) / 1e9 / 8000.0, 'otf': os.environ.get('HELM_MG3_OTF', '1')}))
