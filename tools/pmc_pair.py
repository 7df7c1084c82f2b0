#!/usr/bin/env python3
"""Pair the k_zgemm3 dispatches of a PMC run (tools/pmc_by_dispatch.py ... seq.txt) with the `[gemm log]` lines of the same program (same launch order):
HBM traffic against the operand bytes the library books, per shape, largest excess first.   python tools/pmc_pair.py seq.txt gemm_seq.txt [rows]"""
import collections, re, sys
pat = re.compile(r'\[gemm log\] M (\d+) N (\d+) K (\d+) batch (\d+) mode (\d+) : ([\d.]+) us, ([\d.]+) TFLOP/s, ([\d.]+) GB/s')
logs = [pat.search(l).groups() for l in open(sys.argv[2]) if pat.search(l)]
seq = [l.rstrip('\n').split('\t') for l in open(sys.argv[1])]
assert len(logs) == len(seq), (len(logs), len(seq))
agg = collections.OrderedDict()
for lg, sq in zip(logs, seq):
    M, N, K, b, mode = map(int, lg[:5]); us = float(lg[5]); gbs = float(lg[7])
    e = agg.setdefault((M, N, K, b, mode, sq[1]), [0, 0.0, 0.0, 0.0, 0.0, int(sq[2])])
    e[0] += 1; e[1] += float(sq[3]); e[2] += float(sq[4]); e[3] += gbs * us * 1e-6; e[4] += us
rows = sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2] - kv[1][3]))
tm = sum(v[1] + v[2] for v in agg.values()); to = sum(v[3] for v in agg.values())
print('measured %.1f GB, operand %.1f GB, ratio %.3f' % (tm, to, tm / to))
print('  excess  measured ( read + written)  operand ratio   n  us(log run)     M     N     K  batch mode kernel, workgroups')
for (M, N, K, b, mode, kn), (n, rd, wr, op, us, wg) in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print('%8.2f  %7.2f (%6.2f+%6.2f)  %7.2f %5.2f %3d %7.0f  %5d %5d %5d %6d %d %s %d' % (rd + wr - op, rd + wr, rd, wr, op, (rd + wr) / op if op else 0, n, us, M, N, K, b, mode, kn.replace('k_zgemm3', ''), wg))
