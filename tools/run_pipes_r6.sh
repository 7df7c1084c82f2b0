#!/bin/bash
# Round 6 experiment: several device pipelines on the one GPU (HELM_BENCH_PIPES) against one, fresh process each.   tools/run_pipes_r6.sh <outdir> <reps> "<pipes:group:spare> ..."
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
REPS=${2:-2}
CASES=${3:-"1:2:2 3:1:4 2:1:4"}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for r in $(seq 1 $REPS); do
  for c in $CASES; do
    IFS=: read P G S <<< "$c"
    HELM_BENCH_PIPES=$P HELM_POOL_SPARE=$S HELM_ALLOC_TRACE=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --group $G --no-cpu --no-config5 --no-config2 --no-config4 --no-host-api > $OUT/p${P}g${G}s${S}_$r.json 2> $OUT/p${P}g${G}s${S}_$r.err
    python3 - $OUT/p${P}g${G}s${S}_$r.json $P $G $S <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d['config']
g = [round(b - a, 1) for a, b in zip([0] + d['item_done_ms'][:-1], d['item_done_ms'])]
print('pipes %s group %s spare %s: value %.0f unprofiled %.0f strong %.0f allocs %s %.2f ms maxgap %.1f first-launches %s throttled %.1f' % (sys.argv[2], sys.argv[3], sys.argv[4], d['value'], c['unprofiled_wfs'], c['strong_job_wfs'], c['timed_dev_allocs'], c['timed_dev_alloc_ms'], c['timed_max_item_gap_ms'], c['timed_first_launches'], c['timed_cpu_throttled_ms']))
print('    gaps', g)
PY
  done
done
