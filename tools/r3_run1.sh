cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_direct.py -x -q -m gpu > gpurun_out/r3a/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3a/tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r3a/bench_pipe.json 2> gpurun_out/r3a/bench_pipe.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-pipeline > gpurun_out/r3a/bench_serial.json 2> gpurun_out/r3a/bench_serial.err
tail -3 gpurun_out/r3a/tests.log
python - <<'PY'
import json
for n in ('pipe','serial'):
    try:
        d=json.loads(open('gpurun_out/r3a/bench_%s.json'%n).read().strip().splitlines()[-1])
        print(n, d['value'], d['ms_per_step'], d['unprofiled'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['config']['device_ms_per_step'], d['stencil_roofline']['frac'])
    except Exception as e:
        print(n, 'failed', e)
PY
