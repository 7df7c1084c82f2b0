#!/bin/bash
# config 4's five dpred / Jtvec times when the leg runs inside the whole bench (all GPU legs before it), fresh process each.   tools/c4_in_full_bench.sh [runs]
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${1:-2}); do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > gpurun_out/c4full.json 2>/dev/null
  python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/c4full.json').read().strip().splitlines()[-1])
c = d['detail']['config4'] if 'config4' in d.get('detail', {}) else d['config4']
print('value %.0f  c4 dpred' % d['value'], [round(t * 1e3, 1) for t in c['dpred_seconds_all']], 'jtvec', [round(t * 1e3, 1) for t in c['jtvec_seconds_all']], 'spreads %.2f %.2f' % (c['dpred_spread'], c['jtvec_spread']), 'c5 %.2f' % d['config']['c5_job_s'])
PY
done
