"""Latency of the in-place inversions: python tools/gj_lab.py n [n ...]  (one matrix, helm_debug_inverse_bench; ms per inversion)"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zephyr_amd import _lib
lib = _lib.load()
rng = np.random.default_rng(3)
for n in [int(a) for a in sys.argv[1:]] or [32, 64, 256, 512, 1024]:
    A = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) / np.sqrt(n) + 2 * np.eye(n)
    A = np.ascontiguousarray(A)
    ms = ctypes.c_double()
    rc = lib.helm_debug_inverse_bench(0, n, A.ctypes.data_as(ctypes.c_void_p), 50, 0, ctypes.byref(ms))
    print('n %5d  %8.1f us per inversion  (%.1f us per 32 columns)  rc %d' % (n, 1e3 * ms.value, 1e3 * ms.value / max(1, n // 32), rc), flush=True)
