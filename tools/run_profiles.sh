#!/bin/bash
# Round-2 profile collection on the GPU box (run from the repo root through gpurun):
#   tools/run_profiles.sh <outdir under gpurun_out>
# 1. python bench.py (all CPU legs)                                   -> bench_n1.json
# 2. rocprofv3 --kernel-trace --stats of bench.py --no-cpu             -> stats CSV + the bench line printed under the profiler
# 3. two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of one work item -> HBM bytes per launch of k_zgemm2 and of k_resid_nm
# 4. config 5 (3-D 256 x 256 x 128): tools/bench3d.py JSON + rocprofv3 stats
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-plain-pass > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_reduce.py --fetch $OUT/pmc_FETCH_SIZE --write $OUT/pmc_WRITE_SIZE --kernel k_zgemm2 --out $OUT/pmc_traffic_zgemm.json --note "all k_zgemm2 dispatches of one work item (factorisation + solve passes)" > /dev/null
python3 tools/pmc_reduce.py --fetch $OUT/pmc_FETCH_SIZE --write $OUT/pmc_WRITE_SIZE --kernel k_resid_nm --out $OUT/pmc_traffic_resid.json --note "node-major residual launches (9-point stencil apply + q operand) of one work item" > /dev/null
python3 tools/pmc_reduce.py --fetch $OUT/pmc_FETCH_SIZE --write $OUT/pmc_WRITE_SIZE --kernel k_stencil_t --out $OUT/pmc_traffic_stencil_micro.json --note "rhs-major stencil apply launches of the in-bench microbenchmark (B = 1, 8, 32, 64; 6 launches each)" > /dev/null
# keep only the summaries (the per-dispatch counter CSVs are tens of MB)
find $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE -name "*.csv" -size +2M -delete
HELM_ND_TRACE=1 python3 tools/bench_direct.py --freqs 5.5 > $OUT/trace.txt 2>&1
python3 tools/bench3d.py --freqs 2 3 4 5 --nsrc 16 > $OUT/bench3d.txt 2> $OUT/bench3d.err
# the cycle it replaced on oversampled grids (standard coarsening, weak layer, large shift), 4 sources at the two end frequencies
python3 tools/bench3d.py --no-apply --standard-cycle --freqs 2 5 --nsrc 4 > $OUT/bench3d_standard.txt 2> $OUT/bench3d_standard.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats3d -o s -- python3 $GRAFT_REPO_ROOT/tools/bench3d.py --freqs 5 --nsrc 16 > $OUT/bench3d_under_rocprof.txt 2> $OUT/stats3d.err
find $OUT/stats $OUT/stats3d -name "*kernel_trace.csv" -size +8M -delete
ls -la $OUT
