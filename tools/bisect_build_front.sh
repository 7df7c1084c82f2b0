#!/bin/bash
# round 6: k_nd_build_front, round 4's tree (r4tree/, a git worktree of 2f09bde^) against the current one: the same serial command under the profiler,
# per-launch durations by launch shape
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
F="--no-cpu --no-config5 --no-host-api --no-pipeline --steps 8 --warmup 2 --no-plain-pass --no-roofline-pass"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r4 -o s -- python3 $GRAFT_REPO_ROOT/r4tree/bench.py $F > $OUT/r4.json 2> $OUT/r4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r6 -o s -- python3 $GRAFT_REPO_ROOT/bench.py $F --no-config2 --no-config4 > $OUT/r6.json 2> $OUT/r6.err
python3 - $OUT <<'PY'
import csv, sys, collections, glob, os
out = sys.argv[1]
for tag in ('r4', 'r6'):
    f = glob.glob(os.path.join(out, tag, '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    items = sum(1 for r in rows if 'k_resid_nm_lds' in r['Kernel_Name'])
    by = collections.defaultdict(list)
    tot = collections.defaultdict(float)
    for r in rows:
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        tot[n.split('<')[0]] += d
        if n.startswith('k_nd_build_front'):
            by[(int(r['Grid_Size_X']) // 256, int(r['Grid_Size_Y']), int(r.get('LDS_Block_Size', r.get('LDS_Block_Size_v', 0)) or 0))].append(d)
    print('== %s: %d items; k_nd_build_front %.3f ms / item; all kernels %.3f ms / item' % (tag, items, tot['k_nd_build_front'] / 1e3 / items, sum(tot.values()) / 1e3 / items))
    for k in sorted(by, key=lambda k: -k[1]):
        v = by[k]
        print('     grid (%5d x %6d) lds %6d: %3d launches, avg %8.1f us, min %8.1f, max %8.1f' % (k[0], k[1], k[2], len(v), sum(v) / len(v), min(v), max(v)))
    top = sorted(tot.items(), key=lambda kv: -kv[1])[:12]
    print('     top: ' + ', '.join('%s %.2f' % (n, t / 1e3 / items) for n, t in top))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
