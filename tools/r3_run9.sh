cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
HELM_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu --no-config5 --no-host-api > gpurun_out/r3k/bench_g2_weak.json 2> gpurun_out/r3k/g2w.err; echo "rc $?"
HELM_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --scaling strong --warmup 1 --no-cpu --no-config5 --no-host-api > gpurun_out/r3k/bench_g2_strong.json 2> gpurun_out/r3k/g2s.err; echo "rc $?"
timeout 600 python bench.py --scaling strong --warmup 2 --no-cpu --no-config5 --no-host-api > gpurun_out/r3k/bench_g1_strong.json 2> gpurun_out/r3k/g1s.err; echo "rc $?"
python - <<'PY'
import json
for nme in ('g2_weak','g2_strong','g1_strong'):
    try:
        d=json.loads(open('gpurun_out/r3k/bench_%s.json'%nme).read().strip().splitlines()[-1])
        print(nme, d['value'], d['ms_per_step'], d['n_gpus'], d['steps'], d['scaling'], d['config']['solves_or_iterations_per_rhs_mean'])
    except Exception as e:
        print(nme, 'failed', e)
PY
tail -3 gpurun_out/r3k/g2s.err
