cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/hl
for i in 1 2 3 4 5; do
  HELM_ALLOC_TRACE=1 python bench.py --no-cpu --no-config5 > gpurun_out/hl/b$i.json 2> gpurun_out/hl/b$i.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/hl/b$i.json').read().strip().splitlines()[-1])
for r in d.get('value_host_api_runs'): print($i, r['workers_per_device'], round(r['value']), r['arrivals_ms'])
PY
  grep -v hipHostMalloc gpurun_out/hl/b$i.err | grep "helm alloc" | tail -8
done
