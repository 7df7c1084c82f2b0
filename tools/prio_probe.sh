#!/bin/bash
# Round 6: priority class of the set-up / transfer stream (HELM_SETUP_PRIO, HELM_XFER_PRIO: -1 low, 1 high) against the headline, config 2 and config 4.
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for p in "-1 -1" "1 1" "1 -1"; do
  set -- $p
  HELM_SETUP_PRIO=$1 HELM_XFER_PRIO=$2 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-config5 --no-host-api > gpurun_out/prio.json 2>/dev/null
  python3 - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/prio.json').read().strip().splitlines()[-1])
c = d['config']
c4 = d['detail']['config4'] if 'config4' in d.get('detail', {}) else d['config4']
print('setup prio %s xfer prio %s: value %.0f unprofiled %.0f strong %.0f | c2 device %.0f host api %s | c4 dpred %.1f ms jtvec %.1f ms' % (sys.argv[1], sys.argv[2], d['value'], c['unprofiled_wfs'], c['strong_job_wfs'], c['c2_wfs_device'], d['detail']['flat'].get('c2_wfs_host_api'), 1e3 * c['c4_dpred_s'], 1e3 * c['c4_jtvec_s']))
PY
done
done
