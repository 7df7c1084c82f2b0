cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3l
timeout 900 python -m pytest tests/test_gpu_direct.py tests/test_gpu_layouts.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r3l/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3l/tests.log
tail -5 gpurun_out/r3l/tests.log
for L in 1 0; do
HELM_ND_SCHURGATHER=$L timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-config5 --no-host-api > gpurun_out/r3l/bench_sg_$L.json 2> gpurun_out/r3l/bench_$L.err
done
python - <<'PY'
import json
for nme in ('1','0'):
    try:
        d=json.loads(open('gpurun_out/r3l/bench_sg_%s.json'%nme).read().strip().splitlines()[-1])
        print('schur gather',nme, d['value'], d['ms_per_step'], d['unprofiled'], d['roofline']['frac'], d['config']['solves_or_iterations_per_rhs_mean'])
    except Exception as e:
        print(nme, 'failed', e)
PY
for L in 1 0; do HELM_ND_SCHURGATHER=$L HELM_ND_TRACE=1 python3 tools/bench_direct.py --freqs 5.5 2>&1 | grep "nd trace" | grep -E "factor" | tail -17 | awk '{print $NF, $(NF-1)}' | tr '\n' ' '; echo; done
