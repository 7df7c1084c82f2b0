#!/usr/bin/env python3
"""HIP calls of all threads in the milliseconds before each long no-kernel gap (rocprofv3 --hip-trace --kernel-trace csv).  tools/trace_before_gap.py <dir> [gap ms] [look-back ms]"""
import csv, sys, glob, os, collections
d = sys.argv[1]; thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0; back = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
def load(pat):
    f = glob.glob(os.path.join(d, '**', pat), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
K = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:40]) for r in load('*kernel_trace.csv'))
A = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Function'], r['Thread_Id']) for r in load('*hip_api_trace.csv'))
t00 = K[0][0]; end = K[0][1]
for k in K[1:]:
    if k[0] - end > thr * 1e6:
        g0 = end
        print('--- gap %.2f ms from %.2f ms' % ((k[0] - g0) / 1e6, (g0 - t00) / 1e6))
        c = collections.Counter()
        for s, e, fn, th in A:
            if s > g0 - back * 1e6 and s < g0 + 1e6:
                if fn in ('hipLaunchKernel', 'hipExtLaunchKernel', 'hipGetLastError', 'hipEventRecord', 'hipSetDevice', '__hipPushCallConfiguration', '__hipPopCallConfiguration', 'hipExtModuleLaunchKernel') and (e - s) < 1e6: c[(fn, th)] += 1; continue
                print('   %+8.3f ms  %9.3f ms  %-34s thread %s' % ((s - g0) / 1e6, (e - s) / 1e6, fn, th))
        print('   (plus %s)' % dict(c))
    if k[1] > end: end = k[1]
