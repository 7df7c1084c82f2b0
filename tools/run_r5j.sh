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
    print('%-22s value %8.0f  ms/step %6.2f  unprofiled %8.0f  strong %8.0f' % (sys.argv[2], d['value'], d['ms_per_step'], c.get('unprofiled_wfs') or 0, c.get('strong_job_wfs') or 0))
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
run warm A=1
run default_1 A=1
run look2_1 HELM_BENCH_LOOKAHEAD=2
run default_2 A=1
run look2_2 HELM_BENCH_LOOKAHEAD=2
run look3 HELM_BENCH_LOOKAHEAD=3
run prio0 HELM_PF_PRIO=0
run look2_prio0 HELM_PF_PRIO=0 HELM_BENCH_LOOKAHEAD=2
