#!/bin/bash
# Round 6: the node-major residual launch of the 1024^2 x 256 job, sparse right-hand sides (production) and dense (HELM_ND_SPARSE_RHS=0): rocprofv3 kernel stats of a short
# serial bench run.    tools/resid_probe.sh <outdir>
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $OUT
export TMPDIR=/tmp
for m in 1 0; do
  export HELM_ND_SPARSE_RHS=$m
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/m$m -o t -- python3 bench.py --steps 6 --warmup 2 --no-pipeline --no-cpu --no-config5 --no-config2 --no-config4 --no-host-api --no-plain-pass --no-roofline-pass > $OUT/m$m.json 2> $OUT/m$m.err
  f=$(find $OUT/m$m -name "*kernel_stats.csv" | head -1)
  echo "HELM_ND_SPARSE_RHS=$m"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_resid_nm' in r['Name']:
        print('   %-40s calls %4s avg %8.1f us min %8.1f max %8.1f' % (r['Name'].split('::')[-1][:40], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
  rm -rf $OUT/m$m
done
