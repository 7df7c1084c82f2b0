cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rec
for r in 100000 1536 1000 600; do
  HELM_ND_RECURSE_N=$r python bench.py --no-cpu --no-host-api --steps 4 --warmup 2 > gpurun_out/rec/b$r.json 2> gpurun_out/rec/b$r.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/rec/b$r.json').read().strip().splitlines()[-1])
c=d['config5']
print('recurse_n', $r, 'job', round(c['job_seconds'],3), [(r['freq_hz'], round(r['seconds'],3), round(r['setup_seconds'],3), max(r['iterations'])) for r in c['per_frequency']])
PY
done
