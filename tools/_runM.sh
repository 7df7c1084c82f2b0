cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_m; mkdir -p $OUT
timeout 2000 python -m pytest tests/test_gpu_direct.py tests/test_gpu_stable_fronts.py tests/test_gpu_dispatch.py tests/test_gpu_layouts.py -x -q -m gpu 2>&1 | tail -6
HELM_ND_TRACE=1 python3 tools/bench_direct.py --freqs 5.5 2>&1 | grep 'nd trace' | sed -n 35,51p | awk '{print $3,$4,$5,$(NF-1)}' | tr '\n' ';'
echo
python3 bench.py --no-cpu --no-config5 --no-host-api --steps 16 --warmup 4 > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['driver_visible'])
PY
