cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 900 python bench.py --no-cpu 2> /dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c5 = d['config5']
print('default run $i:', round(d['value']), round(d['ms_per_step'],1), round(d['unprofiled']['value']), 'host', round(d['value_host_api']['value']), [round(r['value']) for r in d['value_host_api_runs']], 'cfg5', round(c5['job_seconds'],2), [round(p['seconds'],2) for p in c5['per_frequency']])
"
done
