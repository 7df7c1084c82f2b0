cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c5p
for p in -1 -1 -1; do
HELM_PF3_PRIO=$p timeout 900 python bench.py --no-cpu --no-host-api --steps 2 --warmup 1 > gpurun_out/c5p/p.json 2> gpurun_out/c5p/p.err
python - <<PY
import json
d=json.loads(open('gpurun_out/c5p/p.json').read().strip().splitlines()[-1])
c=d['config5']
t=c['pipelined_timeline_ms']
sol=[(a[1], round(b[2]-a[2])) for a in t if a[0]=='solve starts' for b in t if b[0]=='solve done' and b[1]==a[1]]
pre=[(a[1], round(b[2]-a[2])) for a in t if a[0]=='prepare starts' for b in t if b[0]=='prepare done' and b[1]==a[1]]
print('prio', $p, 'pipelined', round(c['job_seconds'],3), 'serial', round(c['job_seconds_one_after_the_other'],3), 'solves', sol, 'prepares', pre)
PY
done
