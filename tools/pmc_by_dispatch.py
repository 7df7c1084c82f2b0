#!/usr/bin/env python3
"""HBM traffic of every distinct (kernel, grid) of two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), largest first:
   python tools/pmc_by_dispatch.py <fetch dir> <write dir> [pattern]"""
import collections, csv, glob, os, re, sys


SEQ = {}


def load(path, counter, pattern):
    out = collections.OrderedDict()
    for f in glob.glob(os.path.join(path, '**', '*counter_collection.csv'), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') != counter or not any(pt in r.get('Kernel_Name', '') for pt in pattern.split(',')):
                continue
            mm = re.search(r'(k_\w+(?:<[^>]*>)?)', r['Kernel_Name'])
            name = mm.group(1) if mm else r['Kernel_Name'][:40]
            grid = tuple(int(r.get(k, 0) or 0) for k in ('Grid_Size_X', 'Grid_Size_Y', 'Grid_Size_Z')) if 'Grid_Size_X' in r else (int(r.get('Grid_Size', 0)),)
            key = (int(r['Dispatch_Id']), name, grid)
            per[key] = per.get(key, 0.0) + float(r['Counter_Value'])
        for (did, name, grid), v in sorted(per.items()):
            e = out.setdefault((name, grid), [0, 0.0])
            e[0] += 1; e[1] += v
            SEQ.setdefault(counter, []).append((did, name, grid[0] // 256, v))
    return out


fetch = load(sys.argv[1], 'FETCH_SIZE', sys.argv[3] if len(sys.argv) > 3 else 'k_zgemm3')
write = load(sys.argv[2], 'WRITE_SIZE', sys.argv[3] if len(sys.argv) > 3 else 'k_zgemm3')
rows = []
for key, (n, kib) in fetch.items():
    w = write.get(key, [0, 0.0])
    rows.append((2 * kib * 1024 / 1e9 + w[1] * 1024 / 1e9, 2 * kib * 1024 / 1e9, w[1] * 1024 / 1e9, n, key))
rows.sort(reverse=True)
print('total %.2f GB (read %.2f, written %.2f) over %d launches' % (sum(r[0] for r in rows), sum(r[1] for r in rows), sum(r[2] for r in rows), sum(r[3] for r in rows)))
for t, rd, wr, n, (name, grid) in rows[:60]:
    print('%8.3f GB  read %8.3f  written %8.3f  n %4d  %-40s grid %s' % (t, rd, wr, n, name, grid))

if len(sys.argv) > 4:       # the dispatches in launch order: ordinal, kernel, workgroups, read GB, written GB
    with open(sys.argv[4], 'w') as fo:
        for (d0, n0, g0, v0), (d1, n1, g1, v1) in zip(SEQ.get('FETCH_SIZE', []), SEQ.get('WRITE_SIZE', [])):
            fo.write('%d\t%s\t%d\t%.6f\t%.6f\t%s\n' % (d0, n0, g0, 2 * v0 * 1024 / 1e9, v1 * 1024 / 1e9, 'ok' if (n0, g0) == (n1, g1) else 'MISMATCH ' + n1))
