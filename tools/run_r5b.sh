#!/bin/bash
# round 5, second GPU call: the new tests, headline variants (direct output, one-launch block steps for smaller fronts, two solve threads), full bench line
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_layouts.py tests/test_gpu_3d.py tests/test_gpu_fullsize.py tests/test_gpu_3d_config5.py -m gpu -x -q > $OUT/tests_new.log 2>&1
tail -15 $OUT/tests_new.log
SHORT="--no-cpu --no-config5 --no-host-api --no-config2 --no-config4 --no-roofline-pass --steps 16 --warmup 4"
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python3 bench.py $SHORT > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python3 - "$OUT/bench_$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d['config']
    print('%-22s value %8.0f  ms/step %6.2f  unprofiled %8.0f  dense %8.0f  support %8.0f  strong %8.0f' % (sys.argv[2], d['value'], d['ms_per_step'], c.get('unprofiled_wfs') or 0, c.get('dense_rhs_wfs') or 0, c.get('support_declared_wfs') or 0, c.get('strong_job_wfs') or 0))
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
run default A=1
run directout0 HELM_ND_DIRECT_OUT=0
run default_again A=1
run gjstep256 HELM_ND_GJSTEP_MIN=256
run gjstep128 HELM_ND_GJSTEP_MIN=128
run gjstep64 HELM_ND_GJSTEP_MIN=64
run solvers2 HELM_BENCH_SOLVERS=2
run solvers2_la2 HELM_BENCH_SOLVERS=2 HELM_BENCH_LOOKAHEAD=2
timeout 900 python3 bench.py --no-cpu > $OUT/bench_full.json 2> $OUT/bench_full.err
python3 - $OUT/bench_full.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(json.dumps(d['config'], indent=0)[:3000])
print('config2', d.get('config2')); print('config4', d.get('config4'))
c5 = d.get('config5'); print('config5 apply', c5.get('apply') if isinstance(c5, dict) else c5)
PY
ls -la $OUT
