cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_d; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_direct.py -x -q -m gpu -k "zgemm or inverse" 2>&1 | tail -3
HELM_ND_GEMMV=8 timeout 600 python -m pytest tests/test_gpu_direct.py -x -q -m gpu -k "zgemm or inverse" 2>&1 | tail -3
HELM_GEMM_LOG=1 python3 tools/bench_direct.py --freqs 5.5 > $OUT/bd.json 2> $OUT/gemm_log.txt
python3 tools/gemm_log.py $OUT/gemm_log.txt 30
python3 bench.py --no-cpu --no-config5 --no-host-api --steps 16 --warmup 4 > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['driver_visible'])
PY
