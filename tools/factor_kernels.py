#!/usr/bin/env python3
"""One warm factorisation of the 1024^2 bench operator under `rocprofv3 --kernel-trace`: run this script under the profiler, then `tools/factor_kernels.py --report <dir>`
lists the kernels of the LAST factorisation in launch order (name, grid, microseconds, gap to the previous kernel) -- where a tree level's time goes.
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/fk -o t -- python3 tools/factor_kernels.py [nf]
    python3 tools/factor_kernels.py --report /tmp/fk"""
import os, sys
if len(sys.argv) > 2 and sys.argv[1] == '--report':
    import csv, glob
    f = glob.glob(os.path.join(sys.argv[2], '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = sorted(list(csv.DictReader(open(f))), key=lambda r: int(r['Start_Timestamp']))
    # the last factorisation starts at the last k_assemble_eurus
    last = max(i for i, r in enumerate(rows) if 'k_assemble_eurus' in r['Kernel_Name'])
    prev_end = None; tot = 0.0
    for r in rows[last:]:
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        prev_end = max(prev_end or e, e)
        tot += (e - s) / 1e3
        print('%-44s grid %6d x %5d x %5d  %8.1f us   gap %7.1f us' % (n[:44], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']), (e - s) / 1e3, gap))
    print('sum of kernel durations %.1f us' % tot)
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import zephyr_amd as za
from zephyr_amd.models import marmousi_like
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = 1024
c = marmousi_like(n, n, 9.0).astype(np.complex128)
cfg = dict(nx=n, nz=n, dx=9.0, dz=9.0, c=c, nPML=10, rtol=1e-10, method='direct', batch=256)
for rnd in range(3):
    ops = [za.Eurus(dict(cfg, freq=f + 0.01 * rnd)) for f in [5.5, 7.5, 3.5, 9.5][:nf]]
    for op in ops: op.handle
    torch.cuda.synchronize()
    za.prefactor_many(ops) if nf > 1 else ops[0].prefactor()
    torch.cuda.synchronize()
    for op in ops: del op.factors
