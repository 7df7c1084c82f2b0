#!/usr/bin/env python3
"""Kernel trace (rocprofv3 --kernel-trace, csv) -> the long gaps with no kernel in flight and the long kernels: where a job stalls.
usage: tools/trace_gaps.py <kernel_trace.csv> [threshold ms = 15]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
K = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:50], r['Queue_Id'] + ':' + r['Stream_Id']) for r in rows)
t00 = K[0][0]
print('%d kernels over %.1f ms' % (len(K), (max(k[1] for k in K) - t00) / 1e6))
end = K[0][1]; last = K[0]
for k in K[1:]:
    if k[0] - end > thr * 1e6:
        print('gap of %8.2f ms with NO kernel in flight, from %9.2f ms: after %-40s (queue %s), before %-40s (queue %s)' % ((k[0] - end) / 1e6, (end - t00) / 1e6, last[2], last[3], k[2], k[3]))
    if k[1] > end: end = k[1]; last = k
for k in K:
    if k[1] - k[0] > thr * 1e6: print('kernel of %8.2f ms at %9.2f ms: %s (queue %s)' % ((k[1] - k[0]) / 1e6, (k[0] - t00) / 1e6, k[2], k[3]))
