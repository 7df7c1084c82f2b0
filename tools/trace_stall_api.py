#!/usr/bin/env python3
"""rocprofv3 --hip-trace --hsa-trace --kernel-trace (csv) of a job -> for every gap with no kernel in flight longer than the threshold: the HIP / HSA calls
that were in progress during it (thread, duration).   tools/trace_stall_api.py <dir> [gap ms = 20] [call ms = 3]"""
import csv, sys, glob, os
d = sys.argv[1]; thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0; cthr = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
def load(pat):
    f = glob.glob(os.path.join(d, '**', pat), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
K = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:40]) for r in load('*kernel_trace.csv'))
A = []
for pat, tag in (('*hip_api_trace.csv', 'hip'), ('*hsa_api_trace.csv', 'hsa')):
    for r in load(pat):
        A.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), tag, r['Function'], r['Thread_Id']))
print('%d kernels, %d api calls' % (len(K), len(A)))
t00 = K[0][0]; end = K[0][1]; last = K[0]
for k in K[1:]:
    if k[0] - end > thr * 1e6:
        g0, g1 = end, k[0]
        print('--- gap %.2f ms from %.2f ms (after %s, before %s)' % ((g1 - g0) / 1e6, (g0 - t00) / 1e6, last[2], k[2]))
        for s, e, tag, fn, th in sorted(A):
            if e > g0 and s < g1 and (e - s) > cthr * 1e6:
                print('      %s %-44s thread %s  start %+9.2f ms rel. gap start, duration %8.2f ms' % (tag, fn, th, (s - g0) / 1e6, (e - s) / 1e6))
    if k[1] > end: end = k[1]; last = k
