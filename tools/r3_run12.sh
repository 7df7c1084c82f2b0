cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3n
for i in 1 2; do
timeout 600 python bench.py --steps 32 --warmup 5 --no-cpu --no-config5 --no-host-api > gpurun_out/r3n/bench_$i.json 2> gpurun_out/r3n/bench_$i.err
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-config5 --no-host-api > gpurun_out/r3n/bench_3.json 2> gpurun_out/r3n/bench_3.err
python - <<'PY'
import json
for nme in ('1','2','3'):
    try:
        d=json.loads(open('gpurun_out/r3n/bench_%s.json'%nme).read().strip().splitlines()[-1])
        print(nme, d['value'], d['ms_per_step'], d['unprofiled'], d['roofline']['frac'], d['config']['solves_or_iterations_per_rhs_mean'], d['config']['device_ms_per_step']['solve_call'])
    except Exception as e:
        print(nme, 'failed', e)
PY
