#!/bin/bash
# Round-3 profile collection on the GPU box (run from the repo root through gpurun):
#   tools/run_profiles_r3.sh <outdir under gpurun_out> [githash]
# 1. python bench.py (all legs: pipelined timed region, serial kernel pass, host-API leg, config 5, CPU baselines) -> bench_n1.json
#    and the driver's command line (--steps 20 --warmup 5)                                                        -> bench_driver.json
# 2. rocprofv3 --kernel-trace --stats of the pipelined region and of the serial kernel pass (separate runs)      -> stats CSVs
# 3. two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of one serial work item -> HBM bytes per launch of k_zgemm2 / k_resid_nm_lds / k_stencil_t
# 4. per-level trace of the direct solver, config 5 under rocprofv3
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
echo "${2:-unknown}" > $OUT/githash.txt
python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python3 bench.py --steps 20 --warmup 5 --no-cpu > $OUT/bench_driver.json 2> $OUT/bench_driver.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pipe -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-config5 --no-host-api --steps 8 --warmup 2 --no-plain-pass > $OUT/bench_pipelined_under_rocprof.json 2> $OUT/stats_pipe.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-config5 --no-host-api --no-pipeline --steps 8 --warmup 2 --no-plain-pass > $OUT/bench_serial_under_rocprof.json 2> $OUT/stats_serial.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-config5 --no-host-api --no-pipeline --no-plain-pass > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_reduce.py --fetch $OUT/pmc_FETCH_SIZE --write $OUT/pmc_WRITE_SIZE --kernel k_zgemm2 --out $OUT/pmc_traffic_zgemm.json --note "all k_zgemm2 dispatches of one work item (factorisation + solve passes), serial" > /dev/null
python3 tools/pmc_reduce.py --fetch $OUT/pmc_FETCH_SIZE --write $OUT/pmc_WRITE_SIZE --kernel k_resid_nm --out $OUT/pmc_traffic_resid.json --note "node-major residual launches (9-point stencil apply + q operand + wavefield store) of one work item" > /dev/null
python3 tools/pmc_reduce.py --fetch $OUT/pmc_FETCH_SIZE --write $OUT/pmc_WRITE_SIZE --kernel k_stencil_t --out $OUT/pmc_traffic_stencil_micro.json --note "rhs-major stencil apply launches of the in-bench microbenchmark (B = 1, 8, 32, 64; 6 launches each)" > /dev/null
find $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE -name "*.csv" -size +2M -delete
HELM_ND_TRACE=1 python3 tools/bench_direct.py --freqs 5.5 > $OUT/trace.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats3d -o s -- python3 $GRAFT_REPO_ROOT/tools/bench3d.py --freqs 5 --nsrc 16 > $OUT/bench3d_under_rocprof.txt 2> $OUT/stats3d.err
find $OUT/stats_pipe $OUT/stats_serial $OUT/stats3d -name "*kernel_trace.csv" -size +4M -delete
ls -la $OUT
