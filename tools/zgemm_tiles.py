#!/usr/bin/env python3
"""Tile-shape lab for under-filled launches: time every tile configuration of the v2 kernel on the small-batch shapes of the
upper tree levels (blocked Gauss-Jordan updates, G21, Schur complements)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from zephyr_amd import _lib
lib = _lib.load()
TILES = ['64x64', '32x128', '16x256', '64x32', '32x64', '16x128', '32x32', '16x64']
SHAPES = [  # (label, M, N, K, batch)
    ('GJ update 1024', 1024, 1024, 32, 1), ('GJ update 512 x2', 512, 512, 32, 2), ('GJ update 512 x4', 512, 512, 32, 4),
    ('GJ update 256 x8', 256, 256, 32, 8), ('GJ update 256 x16', 256, 256, 32, 16), ('GJ update 128 x32', 128, 128, 32, 32),
    ('GJ update 128 x64', 128, 128, 32, 64), ('GJ update 64 x128', 64, 64, 32, 128), ('GJ update 64 x256', 64, 64, 32, 256),
    ('GJ update 64 x16384', 64, 64, 32, 16384),
    ('G21 l1', 1024, 512, 512, 2), ('Schur l1', 1024, 1024, 512, 2), ('G21 l2', 1025, 512, 512, 4), ('Schur l2', 1025, 1025, 512, 4),
    ('G21 l3', 1281, 256, 256, 8), ('Schur l3', 1281, 1281, 256, 8), ('G21 l5', 768, 128, 128, 32), ('Schur l5', 768, 768, 128, 32),
    ('G21 l7', 384, 64, 64, 128), ('Schur l7', 384, 384, 64, 128), ('G21 l9', 192, 32, 32, 512), ('Schur l9', 192, 192, 32, 512),
    ('G21 l11', 96, 16, 16, 2048), ('Schur l11', 96, 96, 16, 2048), ('G21 l13', 48, 8, 8, 8192), ('Schur l13', 48, 48, 8, 8192),
    ('rec 512^3', 512, 512, 512, 1), ('rec 256^3', 256, 256, 256, 1), ('rec 128^3', 128, 128, 128, 1), ('rec 64^3', 64, 64, 64, 1),
    ('top bwd s=1024', 1024, 256, 1024, 1), ('l1 fwd', 1024, 256, 512, 2), ('l1 bwd F12', 512, 256, 1024, 2), ('l1 bwd F11', 512, 256, 512, 2),
]
gv = int(sys.argv[1]) if len(sys.argv) > 1 else 1
print('%-20s %5s %5s %5s %6s | ' % ('shape', 'M', 'N', 'K', 'batch') + ' '.join('%-14s' % t for t in ['auto'] + TILES))
rows = []
for label, M, N, K, b in SHAPES:
    fl = 8.0 * M * N * K * b
    reps = max(5, min(50, int(5e10 / fl)))
    cells = []
    for tile in [-1] + list(range(8)):
        ms = ctypes.c_double(0)
        rc = lib.helm_debug_zgemm_bench(0, M, N, K, b, gv + 16 * (tile + 1) if tile >= 0 else gv, reps, ctypes.byref(ms))
        assert rc == 0
        cells.append((ms.value * 1e3, fl / ms.value / 1e9))
    rows.append(dict(label=label, M=M, N=N, K=K, batch=b, us=[c[0] for c in cells]))
    best = min(range(1, 9), key=lambda i: cells[i][0])
    print('%-20s %5d %5d %5d %6d | ' % (label, M, N, K, b) + ' '.join('%7.1f %5.1f%s' % (c[0], c[1], '*' if i == best else ' ') for i, c in enumerate(cells)), flush=True)
print(json.dumps(rows))
