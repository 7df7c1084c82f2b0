#!/usr/bin/env python3
"""Aggregate `[gemm log]` lines (HELM_GEMM_LOG=1, per-launch HIP-event times of the direct solver's products) by shape:
   HELM_GEMM_LOG=1 python tools/bench_direct.py --freqs 5.5 2> log.txt; python tools/gemm_log.py log.txt"""
import collections, re, sys
pat = re.compile(r'\[gemm log\] M (\d+) N (\d+) K (\d+) batch (\d+) mode (\d+) : ([\d.]+) us, ([\d.]+) TFLOP/s, ([\d.]+) GB/s')
agg = collections.OrderedDict()
for line in open(sys.argv[1]):
    m = pat.search(line)
    if not m:
        continue
    M, N, K, b, mode = (int(m.group(i)) for i in range(1, 6))
    us = float(m.group(6))
    e = agg.setdefault((M, N, K, b, mode), [0, 0.0])
    e[0] += 1; e[1] += us
rows = []
for (M, N, K, b, mode), (n, us) in agg.items():
    fl = 8.0 * M * N * K * b
    by = 16.0 * (M * K + K * N + M * N) * b
    rows.append((us, n, M, N, K, b, mode, fl * n / us / 1e6, by * n / us / 1e3, max(fl / 78.6e12, by / 8e12) * 1e6 * n))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('total %.2f ms over %d launches; modes: 0 dense, 1 row table, 2 forward gather, 4 Schur gather, 5 update + pivot sweep' % (tot / 1e3, sum(r[1] for r in rows)))
print('%9s %5s %6s %6s %6s %7s %4s %8s %8s %9s' % ('us total', 'n', 'M', 'N', 'K', 'batch', 'mode', 'TFLOP/s', 'GB/s', 'roof us'))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    print('%9.1f %5d %6d %6d %6d %7d %4d %8.1f %8.0f %9.1f' % r)
