#!/bin/bash
# Round 6 experiment: the headline against the dispatcher's lookahead (HELM_BENCH_LOOKAHEAD) and pool spares, fresh process each.   tools/run_la_r6.sh <outdir> <reps> "<lookahead:spare> ..."
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
REPS=${2:-3}
CASES=${3:-"1:2 2:2 2:3"}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for r in $(seq 1 $REPS); do
  for c in $CASES; do
    IFS=: read L S <<< "$c"
    HELM_BENCH_LOOKAHEAD=$L HELM_POOL_SPARE=$S python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-config5 --no-config2 --no-config4 --no-host-api > $OUT/l${L}s${S}_$r.json 2> $OUT/l${L}s${S}_$r.err
    python3 - $OUT/l${L}s${S}_$r.json $L $S <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d['config']
g = [round(b - a, 1) for a, b in zip([0] + d['item_done_ms'][:-1], d['item_done_ms'])]
print('lookahead %s spare %s: value %.0f unprofiled %.0f strong %.0f allocs %s %.2f ms maxgap %.1f' % (sys.argv[2], sys.argv[3], d['value'], c['unprofiled_wfs'], c['strong_job_wfs'], c['timed_dev_allocs'], c['timed_dev_alloc_ms'], c['timed_max_item_gap_ms']), g)
PY
  done
done
