cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3m
for i in 1 2; do
timeout 600 python bench.py --steps 32 --warmup 5 --no-cpu --no-config5 --no-host-api > gpurun_out/r3m/bench_$i.json 2> gpurun_out/r3m/bench_$i.err
done
HELM_ND_STABLE_THR=5e4 timeout 600 python bench.py --steps 32 --warmup 5 --no-cpu --no-config5 --no-host-api > gpurun_out/r3m/bench_3.json 2> gpurun_out/r3m/bench_3.err
python - <<'PY'
import json
for nme in ('1','2','3'):
    try:
        d=json.loads(open('gpurun_out/r3m/bench_%s.json'%nme).read().strip().splitlines()[-1])
        print(nme, d['value'], d['ms_per_step'], d['unprofiled'], d['roofline']['frac'], d['config']['solves_or_iterations_per_rhs_mean'], d['config']['solves_or_iterations_per_rhs_max'])
    except Exception as e:
        print(nme, 'failed', e)
PY
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
