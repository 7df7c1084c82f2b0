#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for B in 16 8 4; do for M in 1 3 0; do echo -n "apply3d B=$B otf=$M: "; HELM_MG3_OTF=$M timeout 300 python3 tools/apply3d_micro.py $B 2>/dev/null | tail -1 | cut -c1-200; done; done
timeout 1700 python3 -m pytest tests/test_gpu_3d.py tests/test_gpu_layouts.py tests/test_gpu_direct.py tests/test_gpu_parity.py tests/test_gpu_solver.py -m gpu -q > $OUT/tests_new.log 2>&1
tail -12 $OUT/tests_new.log | cut -c1-300
SHORT="--no-cpu --no-config5 --no-host-api --no-config2 --no-config4 --no-roofline-pass --steps 24 --warmup 4"
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
run warm A=1
run default_1 A=1
run gj512_1 HELM_ND_GJSTEP_MIN=512
run default_2 A=1
run gj512_2 HELM_ND_GJSTEP_MIN=512
timeout 600 python3 tools/profile_c4.py > $OUT/profile_c4.txt 2>&1; grep -A40 "^===" $OUT/profile_c4.txt | grep -v "^ *[0-9]\+ \+[0-9.]\+ \+[0-9.]\+ \+[0-9.]\+ \+[0-9.]\+ " | cut -c1-160 | head -90
timeout 900 python3 bench.py --no-cpu --no-host-api > $OUT/bench_full.json 2> $OUT/bench_full.err
python3 - $OUT/bench_full.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['two_roofs']['frac'])
c5 = d.get('config5'); print('config5 job', c5.get('job_seconds'), c5.get('job_seconds_rtol1e10'), 'apply', [(a['B'], round(a['us']), round(a['frac_of_peak'], 3)) for a in c5.get('apply', [])] if isinstance(c5, dict) else c5)
print({k: v for k, v in d['config'].items() if k.startswith('c2_') or k.startswith('c4_') or k.startswith('c5_')})
PY
