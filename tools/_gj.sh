#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gjp -o s -- python3 $GRAFT_REPO_ROOT/tools/gj_lab.py 8 16 32 > /tmp/gj.txt 2>&1
cat /tmp/gj.txt | grep -v amdgpu
f=$(find /tmp/gjp -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r['Kernel_Name'][:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    # three sizes, 51 launches each, in order
    for s in range(0, len(v), 51):
        seg = v[s:s + 51]
        print(k, len(seg), 'avg %.2f us  min %.2f' % (sum(seg) / len(seg), min(seg)))
PY
