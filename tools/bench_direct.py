#!/usr/bin/env python3
"""Time the sparse direct path on the bench workload: factorisation and solve of B sources at one frequency."""
import argparse, json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--grid', type=int, default=1024)
    ap.add_argument('--dx', type=float, default=9.0)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--freqs', type=str, default='2.0,5.5,9.5')
    ap.add_argument('--rtol', type=float, default=1e-10)
    ap.add_argument('--tti', action='store_true', help='coupled TTI system: smooth theta, eps != delta')
    args = ap.parse_args()
    import torch
    import bench
    import zephyr_amd as za
    cfg = bench.build_config(args.grid, args.dx)
    n = args.grid
    if args.tti:
        zz, xx = np.mgrid[0:n, 0:n] / float(n)
        cfg.update(theta=0.3 * np.sin(2 * np.pi * xx) * np.cos(np.pi * zz), eps=0.15 + 0.1 * np.sin(3 * np.pi * zz), delta=0.05 + 0.05 * np.cos(2 * np.pi * xx))
    for f in [float(v) for v in args.freqs.split(',')]:
        sc = dict(cfg, freq=f, method='direct', rtol=args.rtol, batch=args.batch)
        op = za.Eurus(sc)
        op.setProfiling(True)
        src = np.stack([np.linspace(200., args.dx * n - 200., args.batch), np.full(args.batch, 20.)], 1)
        q = za.SparseKaiserSource(sc)(src)
        R = torch.from_numpy(np.ascontiguousarray(q.toarray().T)).cuda()
        U = torch.empty_like(R)
        out = []
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            op.solveDevice(R.data_ptr(), U.data_ptr(), args.batch, n * n)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            t = op.lastTiming()
            info = op.lastInfo
            out.append(dict(wall_s=dt, **t, solves=max(i['iterations'] for i in info), relres=max(i['relres'] for i in info),
                            bad=sum(i['status'] != 0 for i in info)))
        for o in out:
            o['gemm_tflops'] = o['gemm_flops'] / max(o['gemm_ms'], 1e-9) / 1e9
        print(json.dumps(dict(freq=f, grid=n, batch=args.batch, runs=out)))


if __name__ == '__main__':
    main()
