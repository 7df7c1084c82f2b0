cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
timeout 600 python -m pytest tests/test_gpu_direct.py -x -q -m gpu -k "ill_conditioned or matches or random" > gpurun_out/r3j/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3j/tests.log
tail -4 gpurun_out/r3j/tests.log
for cfg in "0 5e4" "1 5e4" "1 1e5"; do
  set -- $cfg
  echo "== HELM_ND_STABLE=$1 THR=$2"
  HELM_ND_STABLE=$1 HELM_ND_STABLE_THR=$2 python3 tools/bench_direct.py --freqs 5.5,9.0,8.0,9.5,7.5 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['runs']
        print('freq %.1f factor_ms %.1f  solve_ms first %.1f then %.1f  solves %d relres %.1e' % (d['freq'], r[0]['factor_ms'], r[0]['solve_ms'], r[2]['solve_ms'], r[2]['solves'], r[2]['relres']))
"
done
for L in 1 0; do
HELM_ND_STABLE=$L timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-config5 --no-host-api > gpurun_out/r3j/bench_stable_$L.json 2> gpurun_out/r3j/bench_$L.err
done
python - <<'PY'
import json
for nme in ('1','0'):
    try:
        d=json.loads(open('gpurun_out/r3j/bench_stable_%s.json'%nme).read().strip().splitlines()[-1])
        print('stable',nme, d['value'], d['ms_per_step'], d['unprofiled'], d['roofline']['frac'], d['config']['solves_or_iterations_per_rhs_mean'])
    except Exception as e:
        print(nme, 'failed', e)
PY
