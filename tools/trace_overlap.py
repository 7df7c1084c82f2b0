#!/usr/bin/env python3
"""Kernel trace (rocprofv3 --kernel-trace, csv) of a pipelined bench run -> how the wall time of the timed region is spent: time with no kernel on
the GPU, with kernels of one stream only, with kernels of both streams; per-kernel time inside the window; per-item period.
usage: tools/trace_overlap.py <kernel_trace.csv> [warmup items to drop = 3]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
K = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0], r['Queue_Id'] + ':' + r['Stream_Id']) for r in rows]
K.sort()
res = [k for k in K if k[2].startswith('k_resid_nm_lds')]
print('kernels', len(K), 'residual launches', len(res))
t0, t1 = res[skip - 1][1], res[-1][1]
items = len(res) - skip
print('window %.3f ms, %d items, %.3f ms per item' % ((t1 - t0) / 1e6, items, (t1 - t0) / 1e6 / items))
W = [k for k in K if k[1] > t0 and k[0] < t1]
ev = []
for s, e, n, q in W:
    s, e = max(s, t0), min(e, t1)
    ev.append((s, 1, q)); ev.append((e, -1, q))
ev.sort()
cnt = collections.Counter(); last = t0; hist = collections.Counter()
for t, d, q in ev:
    nq = sum(1 for v in cnt.values() if v > 0)
    tot = sum(cnt.values())
    hist[(nq, min(tot, 3))] += t - last
    last = t
    cnt[q] += d
hist[(0, 0)] += t1 - last
for k in sorted(hist):
    print('streams busy %d, kernels in flight %s: %8.3f ms per item (%4.1f %%)' % (k[0], ('%d' % k[1]) if k[1] < 3 else '>=3', hist[k] / 1e6 / items, 100.0 * hist[k] / (t1 - t0)))
by = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in W:
    by[n.split('<')[0] if not n.startswith('k_zgemm3') else n][0] += 1; by[n.split('<')[0] if not n.startswith('k_zgemm3') else n][1] += min(e, t1) - max(s, t0)
print('kernel time per item (sum of durations, both streams):')
tot = 0
for n, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:28]:
    print('  %-60s %7.1f launches %8.3f ms' % (n[:60], c / items, d / 1e6 / items)); tot += d
print('  total of all kernels %.3f ms per item' % (sum(d for c, d in by.values()) / 1e6 / items))
qs = collections.defaultdict(int)
for s, e, n, q in W: qs[q] += min(e, t1) - max(s, t0)
for q, d in qs.items(): print('  queue:stream %s busy %.3f ms per item' % (q, d / 1e6 / items))
# ---- how full is the chip: workgroups of the kernels in flight (launch sizes, not residency) ----
G = []
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if e <= t0 or s >= t1: continue
    wg = 1
    for ax in 'XYZ':
        wg *= max(1, int(r['Grid_Size_' + ax]) // max(1, int(r['Workgroup_Size_' + ax])))
    G.append((max(s, t0), min(e, t1), wg))
ev = []
for s, e, wg in G: ev.append((s, wg)); ev.append((e, -wg))
ev.sort()
cur = 0; last = t0; fill = collections.Counter()
for t, d in ev:
    b = 0 if cur == 0 else (1 if cur < 64 else (2 if cur < 256 else (3 if cur < 1024 else 4)))
    fill[b] += t - last; last = t; cur += d
names = ['no kernel', '< 64 workgroups launched', '64-255', '256-1023', '>= 1024']
for b in range(5):
    print('in flight: %-26s %8.3f ms per item (%4.1f %%)' % (names[b], fill[b] / 1e6 / items, 100.0 * fill[b] / (t1 - t0)))
