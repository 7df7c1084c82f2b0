#!/usr/bin/env python3
"""Mid-fill launches (a few hundred 64 x 64 tiles): the default tile choice against every forced tile."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zephyr_amd import _lib
lib = _lib.load()
SHAPES_TOP = [(1025, 256, 512, 4), (1281, 256, 256, 8), (1025, 512, 512, 4), (512, 256, 1025, 4), (256, 256, 1281, 8), (512, 256, 1024, 2), (1024, 256, 1024, 1),
          (1024, 256, 512, 2), (768, 256, 128, 32), (384, 256, 64, 128), (128, 256, 768, 32), (1024, 512, 512, 2), (1025, 1025, 512, 4), (1281, 1281, 256, 8)]
SHAPES_LOW = [(8, 256, 48, 8192), (16, 256, 64, 4096), (16, 256, 96, 2048), (32, 256, 128, 1024), (32, 256, 192, 512), (8, 256, 8, 8192), (16, 256, 16, 4096),
              (32, 256, 32, 1024), (32, 256, 49, 16129), (64, 256, 256, 256), (64, 256, 384, 128), (48, 256, 8, 8192), (64, 256, 16, 4096), (96, 256, 16, 2048), (128, 256, 32, 1024)]
SHAPES = SHAPES_LOW if (len(sys.argv) > 1 and sys.argv[1] == 'low') else SHAPES_TOP
vs = [7] + [7 + 16 * (t + 1) for t in range(8)]
print('%-22s | default  ' % 'M N K batch' + ' '.join('tile%d   ' % t for t in range(8)))
for M, N, K, b in SHAPES:
    fl = 8.0 * M * N * K * b
    cells = []
    for v in vs:
        ms = ctypes.c_double(0)
        rc = lib.helm_debug_zgemm_bench(0, M, N, K, b, v, 30, ctypes.byref(ms))
        cells.append(ms.value * 1e3 if rc == 0 else float('nan'))
    print('%5d %4d %5d %4d | ' % (M, N, K, b) + ' '.join('%7.1f' % c for c in cells) + '   us;  best %.1f TF/s' % (fl / min(cells) / 1e6), flush=True)
