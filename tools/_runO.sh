cd $GRAFT_REPO_ROOT
for la in 1 2 3; do
HELM_BENCH_LOOKAHEAD=$la python3 bench.py --no-cpu --no-config5 --no-host-api --steps 16 --warmup 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lookahead $la', round(d['value']), round(d['ms_per_step'],2), round(d['unprofiled']['value']))"
done
