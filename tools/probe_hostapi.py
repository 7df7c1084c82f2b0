"""where the host-array path spends its time (one frequency of the 1024^2 job)"""
import os, sys, time, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g; g.build()
from zephyr_amd import Eurus, SparseKaiserSource, _lib
from bench import build_config, source_locations
n, dx = 1024, 9.0
cfg = build_config(n, dx); N = n * n
q = SparseKaiserSource(cfg)(source_locations(n, dx, 256))
lib = _lib.load()
def T(label, fn, reps=3):
    for r in range(reps):
        t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); print('%-40s %8.1f ms' % (label, 1e3 * (time.perf_counter() - t0)), flush=True)
    return out
nb = N * 256 * 16
blk = T('helm_host_alloc 4.3 GB (+free)', lambda: _lib.pinned_empty((N, 256)))
U = _lib.pinned_empty((N, 256))
d = torch.empty((N, 256), dtype=torch.complex128, device='cuda')
torch.cuda.synchronize()
hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
T('hipMemcpy D2H into pinned', lambda: hip.hipMemcpy(U.ctypes.data, d.data_ptr(), nb, 2))
P = np.empty((N, 256), complex)
T('hipMemcpy D2H into pageable', lambda: hip.hipMemcpy(P.ctypes.data, d.data_ptr(), nb, 2), reps=2)
sc = dict(cfg); sc.update(freq=6.0, rtol=1e-10, method='auto', batch=256)
op = Eurus(sc)
T('op * q (sparse; first = assemble+factor)', lambda: op * q)
T('coo conversion', lambda: q.tocoo())

import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); u = op * q; pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(12)
