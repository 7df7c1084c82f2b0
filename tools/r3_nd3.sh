cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/nd3
run() {
  tag=$1; shift
  env "$@" HELM_MG3_TRACE=1 timeout 900 python bench.py --no-cpu --no-host-api --steps 2 --warmup 1 > gpurun_out/nd3/$tag.json 2> gpurun_out/nd3/$tag.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/nd3/$tag.json').read().strip().splitlines()[-1])
    c=d['config5']
    print('$tag', 'job', round(c['job_seconds'],3) if isinstance(c,dict) and 'job_seconds' in c else c, [(r['freq_hz'], round(r['seconds'],3), round(r['setup_seconds'],3), max(r['iterations'])) for r in c['per_frequency']] if isinstance(c,dict) else '')
except Exception as e:
    print('$tag failed', e)
PY
  grep "column dissection\|mg3 depth" gpurun_out/nd3/$tag.err | head -8
}
run bt HELM_MG3_COARSE=bt
run nd HELM_MG3_COARSE=nd
run nd_rule HELM_MG3_COARSE=nd HELM_MG3_DEPTH_MODEL=0
