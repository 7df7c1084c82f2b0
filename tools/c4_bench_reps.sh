#!/bin/bash
# config 4 through bench.py, fresh process each: the five dpred / Jtvec times of every run.   tools/c4_bench_reps.sh [runs]
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${1:-3}); do
  python3 bench.py --steps 4 --warmup 3 --no-cpu --no-config5 --no-config2 --no-host-api --no-plain-pass --no-roofline-pass > gpurun_out/c4q.json 2>/dev/null
  python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/c4q.json').read().strip().splitlines()[-1])
c = d['detail']['config4'] if 'config4' in d.get('detail', {}) else d['config4']
print('c4 dpred', [round(t * 1e3, 1) for t in c['dpred_seconds_all']], 'jtvec', [round(t * 1e3, 1) for t in c['jtvec_seconds_all']], 'spreads %.2f %.2f' % (c['dpred_spread'], c['jtvec_spread']))
PY
done
