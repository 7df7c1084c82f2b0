#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_gpu_layouts.py -m gpu -q -k direct_output > $OUT/tests_directout.log 2>&1
grep -E "AssertionError|passed|failed" $OUT/tests_directout.log | cut -c1-1500 | head -12
timeout 600 python3 tools/profile_c4.py > $OUT/profile_c4.txt 2>&1; grep -A40 "^===" $OUT/profile_c4.txt | grep -v "^ *[0-9]\+ \+[0-9.]\+ \+[0-9.]\+ \+[0-9.]\+ \+[0-9.]\+ " | grep "calls\|===" | cut -c1-160 | head -60
timeout 1700 python3 -m pytest tests -m gpu -q -x --deselect tests/test_gpu_layouts.py::test_direct_output_writes_the_wavefields_of_the_two_step_path > $OUT/tests_all.log 2>&1
tail -8 $OUT/tests_all.log | cut -c1-300
timeout 900 python3 bench.py --no-cpu --no-host-api > $OUT/bench_full.json 2> $OUT/bench_full.err
python3 - $OUT/bench_full.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['two_roofs']['frac'])
c5 = d.get('config5'); print('config5 job', c5.get('job_seconds'), c5.get('job_seconds_rtol1e10'), 'apply', [(a['B'], round(a['us']), round(a['frac_of_peak'], 3)) for a in c5.get('apply', [])] if isinstance(c5, dict) else c5)
print({k: v for k, v in d['config'].items() if k.startswith('c2_') or k.startswith('c4_') or k.startswith('c5_')})
PY
