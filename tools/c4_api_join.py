#!/usr/bin/env python3
"""HIP / HSA API calls longer than a threshold inside the first part of each dpred call of tools/c4_timeline.py (rocprofv3 --hip-trace --hsa-trace, csv).
   python3 tools/c4_api_join.py <timeline.txt> <trace dir> [min ms]"""
import csv, glob, re, sys
tl, d = sys.argv[1], sys.argv[2]
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
calls = [(int(m.group(1)), float(m.group(2)), int(m.group(3))) for m in re.finditer(r'--- dpred call (\d+): ([\d.]+) ms\s+\(starts at CLOCK_MONOTONIC (\d+) ns\)', open(tl).read())]
ev = []
for pat in ('*hip_api_trace.csv', '*hsa_api_trace.csv'):
    for f in glob.glob(d + '/**/' + pat, recursive=True):
        for r in csv.DictReader(open(f)):
            a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            if b - a >= thr * 1e6: ev.append((a, b, r['Function'], r['Thread_Id']))
ev.sort()
print('%d api calls >= %.1f ms' % (len(ev), thr))
for c, ms, t0 in calls:
    print('--- dpred call %d: %.1f ms' % (c, ms))
    for a, b, fn, th in ev:
        if a >= t0 and a <= t0 + 35e6: print('   %8.3f .. %8.3f (%7.3f ms)  %-40s thread %s' % ((a - t0) / 1e6, (b - t0) / 1e6, (b - a) / 1e6, fn, th))
