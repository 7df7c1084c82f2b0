#!/bin/bash
# Round 6: is the driver's headline reproducible?  The driver's command as the FIRST GPU process of a fresh box, then again in fresh processes on the
# same box; runtime-object counts of the timed region and first-launch / allocation traces on stderr.   tools/run_repro_r6.sh <outdir> [reps]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
REPS=${2:-3}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for i in $(seq 1 $REPS); do
  HELM_LAUNCH_TRACE=1 HELM_ALLOC_TRACE=2 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-config5 --no-config2 --no-config4 --no-host-api ${BENCH_EXTRA:-} > $OUT/run$i.json 2> $OUT/run$i.err
  python3 - $OUT/run$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d['config']
print(sys.argv[1].split('/')[-1], 'value %.0f' % d['value'], 'ms/step %.2f' % d['ms_per_step'], {k: c[k] for k in c if k.startswith('timed_') or k in ('unprofiled_wfs', 'strong_job_wfs')})
print('   item gaps', [round(b - a, 1) for a, b in zip([0] + d['item_done_ms'][:-1], d['item_done_ms'])])
print('   ', {k: v for k, v in d['detail']['flat'].items() if k.startswith(('timed_', 'warm', 'kernels', 'headline', 'weak'))})
PY
done
