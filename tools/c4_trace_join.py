#!/usr/bin/env python3
"""Join tools/c4_timeline.py's host marks with a rocprofv3 kernel + memory-copy trace of the same run: what the GPU did in the first 30 ms of each dpred call.
   python3 tools/c4_trace_join.py <timeline.txt> <trace dir>"""
import csv, glob, re, sys
tl, d = sys.argv[1], sys.argv[2]
calls = [(int(m.group(1)), float(m.group(2)), int(m.group(3))) for m in re.finditer(r'--- dpred call (\d+): ([\d.]+) ms\s+\(starts at CLOCK_MONOTONIC (\d+) ns\)', open(tl).read())]
ev = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K q%s %s' % (r.get('Queue_Id', '?'), r['Kernel_Name'].split('(')[0][-50:])))
for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY %s %s B' % (r.get('Direction', ''), r.get('Size', r.get('Bytes', '?')))))
ev.sort()
print('%d events; trace spans %d .. %d' % (len(ev), ev[0][0], ev[-1][1]))
for c, ms, t0 in calls:
    print('--- dpred call %d: %.1f ms' % (c, ms))
    n = 0
    for a, b, w in ev:
        if a >= t0 - 2e6 and a <= t0 + 32e6:
            print('   %8.3f .. %8.3f (%7.3f ms)  %s' % ((a - t0) / 1e6, (b - t0) / 1e6, (b - a) / 1e6, w)); n += 1
            if n > 40: break
