cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bl
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  HELM_ALLOC_TRACE=1 python bench.py --no-cpu --no-config5 --no-host-api $BENCH_ARGS > gpurun_out/bl/b$i.json 2> gpurun_out/bl/b$i.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/bl/b$i.json').read().strip().splitlines()[-1])
print($i, round(d['value']), round(d['ms_per_step'],2), round(d['unprofiled']['value']), d['item_done_ms'])
PY
  grep "helm alloc" gpurun_out/bl/b$i.err | awk '$NF=="ms" && $(NF-1) > 20' | tail -12
done
