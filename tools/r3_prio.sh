cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prio
for p in 1 0 -1 1 0 -1; do
  HELM_PF_PRIO=$p python bench.py --no-cpu --no-config5 --no-host-api --steps 20 --warmup 5 > gpurun_out/prio/b.json 2> gpurun_out/prio/b.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/prio/b.json').read().strip().splitlines()[-1])
print('prio', $p, round(d['value']), round(d['ms_per_step'],2), round(d['unprofiled']['value']))
PY
done
