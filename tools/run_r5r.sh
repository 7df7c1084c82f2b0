#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_stable_fronts.py tests/test_gpu_direct.py -m gpu -q -x > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log | cut -c1-300
SHORT="--no-cpu --no-config5 --no-host-api --no-config2 --no-config4 --no-roofline-pass --steps 24 --warmup 4"
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python3 bench.py $SHORT > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python3 - "$OUT/bench_$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d['config']
    print('%-22s value %8.0f  ms/step %6.2f  unprofiled %8.0f  strong %8.0f  passes %s' % (sys.argv[2], d['value'], d['ms_per_step'], c.get('unprofiled_wfs') or 0, c.get('strong_job_wfs') or 0, c.get('solves_or_iterations_per_rhs_max')))
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
run warm A=1
run d1 A=1
run d2 A=1
run d3 A=1
F=2.0,2.5,3.0,3.5,4.0,4.5,5.0,5.5,6.0,6.5,7.0,7.5,8.0,8.5,9.0,9.5
HELM_ND_DEBUG=1 timeout 600 python3 tools/bench_direct.py --freqs $F 2>&1 | grep -E "re-eliminated|pass 2" | sort | uniq -c | sort -rn | head -30
