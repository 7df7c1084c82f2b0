cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests/test_gpu_dispatch.py tests/test_gpu_3d_config5.py tests/test_gpu_direct.py -x -q -m gpu -s > gpurun_out/r3b/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3b/tests.log
HELM_MG3_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err
tail -25 gpurun_out/r3b/tests.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3b/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['unprofiled'], d['roofline']['frac'], d['roofline'].get('in_pipeline',{}).get('frac'), d['stencil_roofline']['frac'])
print(json.dumps(d.get('config5'))[:3000])
PY
grep "mg3 depth" gpurun_out/r3b/bench.err | head
