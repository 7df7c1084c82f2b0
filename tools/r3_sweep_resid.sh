cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
for cfgs in "4 1" "2 1" "4 0" "2 0"; do
  set -- $cfgs
  HELM_ND_RESID_RPT=$1 HELM_ND_RESID_LDS=$2 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu --no-config5 --no-host-api --no-pipeline --no-plain-pass > gpurun_out/r3e/c_$1_$2.json 2>/dev/null
  python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r3e/c_%s_%s.json'%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
s=d['stencil_roofline']
print('RPT',sys.argv[1],'LDS',sys.argv[2],'frac %.3f'%s['frac'],'us %.0f'%s['avg_launch_us'],'value %.0f'%d['value'])
PY
done
timeout 300 python -m pytest tests/test_gpu_layouts.py tests/test_gpu_direct.py -x -q -m gpu 2>&1 | tail -3
