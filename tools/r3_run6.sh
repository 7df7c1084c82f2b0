cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
timeout 900 python -m pytest tests/test_gpu_direct.py tests/test_gpu_layouts.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r3h/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3h/tests.log
tail -5 gpurun_out/r3h/tests.log
HELM_ND_DEBUG=1 timeout 300 python3 tools/bench_direct.py --freqs 8.0,9.0,9.5,6.0,7.5,2.0,4.5 2>&1 | grep -E "ill-conditioned|pass 1|pass 2" | awk '{print substr($0,1,150)}' | head -60
for L in 1 0; do
HELM_ND_STABLE=$L timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-config5 --no-host-api > gpurun_out/r3h/bench_stable_$L.json 2> gpurun_out/r3h/bench_$L.err
done
python - <<'PY'
import json
for nme in ('1','0'):
    try:
        d=json.loads(open('gpurun_out/r3h/bench_stable_%s.json'%nme).read().strip().splitlines()[-1])
        print('stable',nme, d['value'], d['ms_per_step'], d['unprofiled'], d['roofline']['frac'], d['config']['solves_or_iterations_per_rhs_mean'], d['config']['device_ms_per_step'])
    except Exception as e:
        print(nme, 'failed', e)
PY
