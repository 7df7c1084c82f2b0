cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests/test_gpu_layouts.py tests/test_gpu_dispatch.py tests/test_gpu_solver.py tests/test_omega.py -x -q -m gpu > gpurun_out/r3d/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3d/tests.log
tail -8 gpurun_out/r3d/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-config5 > gpurun_out/r3d/bench_node.json 2> gpurun_out/r3d/bench_node.err
python - <<'PY'
import json
for nme in ('node',):
    try:
        d=json.loads(open('gpurun_out/r3d/bench_%s.json'%nme).read().strip().splitlines()[-1])
        print(nme, d['value'], d['ms_per_step'], d['unprofiled'], d['roofline']['frac'], d['stencil_roofline']['frac'])
        print(json.dumps(d.get('value_host_api')), json.dumps(d.get('value_host_api_runs')))
    except Exception as e:
        print(nme, 'failed', e)
PY
tail -5 gpurun_out/r3d/bench_node.err
