cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
timeout 900 python -m pytest tests/test_gpu_layouts.py tests/test_gpu_dispatch.py tests/test_gpu_solver.py tests/test_gpu_direct.py -x -q -m gpu > gpurun_out/r3c/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3c/tests.log
tail -25 gpurun_out/r3c/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-config5 > gpurun_out/r3c/bench_node.json 2> gpurun_out/r3c/bench_node.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-config5 --no-host-api --layout rhs > gpurun_out/r3c/bench_rhs.json 2> gpurun_out/r3c/bench_rhs.err
python - <<'PY'
import json
for nme in ('node','rhs'):
    try:
        d=json.loads(open('gpurun_out/r3c/bench_%s.json'%nme).read().strip().splitlines()[-1])
        print(nme, d['value'], d['ms_per_step'], d['unprofiled'], d['roofline']['frac'], d['stencil_roofline']['frac'], d['stencil_roofline']['avg_launch_us'], d['config']['solves_or_iterations_per_rhs_mean'])
        print(json.dumps(d.get('value_host_api')), json.dumps(d.get('value_host_api_runs')))
    except Exception as e:
        print(nme, 'failed', e)
PY
tail -5 gpurun_out/r3c/bench_node.err
