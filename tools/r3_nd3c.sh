cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/nd3
for leaf in 2 3 4; do
HELM_MG3_ND_LEAF=$leaf HELM_MG3_TRACE=1 timeout 900 python bench.py --no-cpu --no-host-api --steps 2 --warmup 1 > gpurun_out/nd3/leaf$leaf.json 2> gpurun_out/nd3/leaf$leaf.err
python - <<PY
import json
d=json.loads(open('gpurun_out/nd3/leaf$leaf.json').read().strip().splitlines()[-1])
c=d['config5']
print('leaf', $leaf, 'job', round(c['job_seconds'],3), [(r['freq_hz'], round(r['seconds'],3), round(r['setup_seconds'],3), round(r['seconds_reusing_setup'],3), max(r['iterations'])) for r in c['per_frequency']])
PY
grep "column dissection" gpurun_out/nd3/leaf$leaf.err | sort | uniq -c | head -3
done
HELM_ND_TRACE=1 timeout 600 python tools/bench3d.py --freqs 5 --nsrc 16 2>&1 | grep "nd trace" | grep "total\|level  [0-3] \|level 1[0-3]" | head -40
