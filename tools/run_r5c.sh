#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_3d.py tests/test_gpu_layouts.py tests/test_gpu_3d_config5.py tests/test_gpu_fullsize.py -m gpu -q > $OUT/tests_new.log 2>&1
tail -25 $OUT/tests_new.log
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
for rep in 1 2; do
run default_$rep A=1
run directout0_$rep HELM_ND_DIRECT_OUT=0
run gjstep128_$rep HELM_ND_GJSTEP_MIN=128
run gjstep256_$rep HELM_ND_GJSTEP_MIN=256
run bigtile9_$rep HELM_ND_BIGTILE=9
run big9_gj128_$rep HELM_ND_BIGTILE=9 HELM_ND_GJSTEP_MIN=128
done
timeout 600 python3 tools/zgemm_lab.py -1,16,160 Schur 20 > $OUT/zgemm_lab.txt 2>&1; cat $OUT/zgemm_lab.txt | grep -v "^\[" | head -20
timeout 600 python3 tools/zgemm_lab.py -1,16,160 "3d plane" 5 >> $OUT/zgemm_lab.txt 2>&1; tail -3 $OUT/zgemm_lab.txt | grep -v "^\[" | cut -c1-200
timeout 600 python3 tools/profile_c4.py > $OUT/profile_c4.txt 2>&1; grep -A22 "^===" $OUT/profile_c4.txt | grep -v "^ *[0-9]* *[0-9.]* *[0-9.]* *[0-9.]* *[0-9.]* " | head -70
timeout 900 python3 bench.py --no-cpu --no-host-api > $OUT/bench_full.json 2> $OUT/bench_full.err
python3 - $OUT/bench_full.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value', d['value'], d['ms_per_step'])
c5 = d.get('config5'); print('config5 job', c5.get('job_seconds'), c5.get('job_seconds_rtol1e10'), 'apply', [(a['B'], round(a['us']), round(a['frac_of_peak'], 3)) for a in c5.get('apply', [])] if isinstance(c5, dict) else c5)
print({k: v for k, v in d['config'].items() if k.startswith('c2_') or k.startswith('c4_') or k.startswith('c5_')})
PY
