cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/nd3
timeout 1500 python -m pytest tests/test_gpu_3d_config5.py tests/test_gpu_3d.py -x -q > gpurun_out/nd3/tests.log 2>&1; tail -5 gpurun_out/nd3/tests.log
HELM_MG3_TRACE=1 timeout 900 python bench.py --no-cpu --no-host-api --steps 2 --warmup 1 > gpurun_out/nd3/auto.json 2> gpurun_out/nd3/auto.err
python - <<PY
import json
d=json.loads(open('gpurun_out/nd3/auto.json').read().strip().splitlines()[-1])
c=d['config5']
print('auto', 'job', round(c['job_seconds'],3), [(r['freq_hz'], round(r['seconds'],3), round(r['setup_seconds'],3), max(r['iterations'])) for r in c['per_frequency']])
PY
grep "column dissection\|mg3 depth" gpurun_out/nd3/auto.err | head -12
