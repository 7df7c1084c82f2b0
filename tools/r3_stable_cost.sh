cd $GRAFT_REPO_ROOT
for cfg in "1 2e4" "1 5e4" "1 1e4"; do
  set -- $cfg
  echo "== HELM_ND_STABLE=$1 THR=$2"
  HELM_ND_STABLE=$1 HELM_ND_STABLE_THR=$2 HELM_ND_DEBUG=1 python3 tools/bench_direct.py --freqs 5.5,9.0,8.0,9.5,7.5 2>&1 | python3 -c "
import sys, json
nfl = 0
for l in sys.stdin:
    if 'ill-conditioned' in l:
        nfl += int(l.split('):')[1].split()[0])
    if l.startswith('{'):
        d = json.loads(l); r = d['runs']
        print('freq %.1f flagged(total over 1 factorisation) %d  factor_ms %.1f  solve_ms first %.1f then %.1f  solves %d relres %.1e' % (d['freq'], nfl, r[0]['factor_ms'], r[0]['solve_ms'], r[2]['solve_ms'], r[2]['solves'], r[2]['relres']))
        nfl = 0
"
done
