cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c5p
for i in 1 2; do
HELM_ALLOC_TRACE=1 timeout 900 python bench.py --no-cpu --no-host-api --steps 2 --warmup 1 > gpurun_out/c5p/b$i.json 2> gpurun_out/c5p/b$i.err
python - <<PY
import json
d=json.loads(open('gpurun_out/c5p/b$i.json').read().strip().splitlines()[-1])
c=d['config5']
print(c if not isinstance(c,dict) else ('pipelined', round(c['job_seconds'],3), 'serial', round(c['job_seconds_one_after_the_other'],3), c['pipelined'], [(r['freq_hz'], round(r['seconds'],3), round(r['setup_seconds'],3), max(r['iterations'])) for r in c['per_frequency']]))
PY
grep "helm alloc" gpurun_out/c5p/b$i.err | awk '$(NF-1) > 20' | tail -12
done
