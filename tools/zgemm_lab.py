#!/usr/bin/env python3
"""Tile-kernel lab: TFLOP/s of the strided-batched complex GEMM (k_zgemm3) on the shapes the 1024^2 plan issues.
usage: zgemm_lab.py [variants] [label filter] [max reps]; variant -1 = the tile the library chooses, 16 (t + 1) = tile configuration t forced
(0 64x64, 1 32x128, 2 16x256, 3 64x32, 4 32x64, 5 16x128, 6 32x32, 7 16x64, 9 128x64; include/helm.h helm_debug_zgemm_bench)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from zephyr_amd import _lib
lib = _lib.load()

SHAPES = [  # (label, M, N, K, batch)
    ('real leaf bwd', 49, 256, 81, 16129), ('real leaf fwd', 32, 256, 49, 16129), ('real leaf G21', 32, 49, 49, 16129), ('real leaf G', 49, 32, 49, 16129),
    ('leaf fwd   G21 x', 32, 256, 64, 16384), ('leaf bwd   F12 x', 64, 256, 32, 16384), ('leaf bwd   F11i t', 64, 256, 64, 16384),
    ('s8  fwd', 48, 256, 8, 8192), ('s8  bwd', 8, 256, 48, 8192),
    ('s16 fwd', 64, 256, 16, 4096), ('s16 bwd', 16, 256, 64, 4096),
    ('s32 fwd', 128, 256, 32, 1024), ('s32 bwd', 32, 256, 128, 1024),
    ('s128 fwd', 512, 256, 128, 64), ('s128 bwd', 128, 256, 512, 64),
    ('s512 fwd', 1025, 256, 512, 4), ('s512 bwd', 512, 256, 1025, 4),
    ('leaf G21 = F21 F11i', 32, 64, 64, 16384), ('leaf Schur', 32, 32, 64, 16384), ('leaf inv 32^3', 32, 32, 32, 16384),
    ('s256 Schur', 1024, 1024, 256, 16), ('s256 G21', 1024, 256, 256, 16),
    ('s256 Schur x8', 1281, 1281, 256, 8), ('s128 Schur x32', 768, 768, 128, 32), ('s128 Schur x64', 512, 512, 128, 64), ('s64 Schur x128', 384, 384, 64, 128),
    ('3d plane 3713 x1', 3713, 3713, 1856, 1),
    ('top inv 512^3 x1', 512, 512, 512, 1), ('top inv 256^3 x2', 256, 256, 256, 2), ('inv 128^3 x8', 128, 128, 128, 8),
    ('inv 64^3 x16', 64, 64, 64, 16), ('inv 32^3 x4', 32, 32, 32, 4),
]
variants = [int(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else '-1'.split(','))]
only = sys.argv[2] if len(sys.argv) > 2 else ''      # substring filter on the shape label
maxreps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
rows = []
print('%-22s %6s %5s %5s %6s | ' % ('shape', 'M', 'N', 'K', 'batch') + ' '.join('%s TF/s (us)   ' % ('auto' if v < 16 else 'tile%d' % (v // 16 - 1)) for v in variants))
for label, M, N, K, b in SHAPES:
    if only and only not in label:
        continue
    fl = 8.0 * M * N * K * b
    reps = max(1, min(maxreps, int(2e11 / fl)))
    cells = []
    for v in variants:
        ms = ctypes.c_double(0)
        rc = lib.helm_debug_zgemm_bench(0, M, N, K, b, v, reps, ctypes.byref(ms))
        assert rc == 0, rc
        cells.append((fl / ms.value / 1e9, ms.value * 1e3))
    rows.append(dict(label=label, M=M, N=N, K=K, batch=b, tflops={str(v): c[0] for v, c in zip(variants, cells)}, us={str(v): c[1] for v, c in zip(variants, cells)}))
    print('%-22s %6d %5d %5d %6d | ' % (label, M, N, K, b) + ' '.join('%6.1f (%7.1f)' % c for c in cells), flush=True)
print(json.dumps(rows))
