#!/bin/bash
# Round-6 profile collection on the GPU box (run from the repo root through gpurun):  tools/run_profiles_r6.sh <outdir under gpurun_out> [githash]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
echo "${2:-unknown}" > $OUT/githash.txt
# the driver's command, as the FIRST GPU process of this box
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
# the headline alone, fresh processes one after the other (the other leases of profiles/r06_headline_repro.json come from separate gpurun calls)
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-config5 --no-config2 --no-config4 --no-host-api > $OUT/headline_$i.json 2> /dev/null; done
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --no-cpu --no-config5 --no-config2 --no-config4 --no-host-api --steps 8 --warmup 3 --no-plain-pass"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pipe -o s -- python3 $B --no-roofline-pass > $OUT/bench_pipelined_under_rocprof.json 2> $OUT/stats_pipe.err
python3 $GRAFT_REPO_ROOT/tools/trace_overlap.py $(find $OUT/stats_pipe -name "*kernel_trace.csv" | head -1) 3 > $OUT/pipeline_overlap.txt 2>&1
HELM_ND_SPARSE_RHS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -o s -- python3 $B --no-pipeline > $OUT/bench_serial_under_rocprof.json 2> $OUT/stats_serial.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial_sparse -o s -- python3 $B --no-pipeline --no-roofline-pass > $OUT/bench_serial_sparse_under_rocprof.json 2> $OUT/stats_serial_sparse.err
# the production launches with nothing beside them: sets of two factorisations, then their solves (the pass `roofline` quotes)
HELM_ND_SPARSE_RHS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_sets -o s -- python3 $B --steps 8 --warmup 4 > $OUT/bench_sets_under_rocprof.json 2> $OUT/stats_sets.err
export HELM_ND_SPARSE_RHS=0
for C in FETCH_SIZE WRITE_SIZE; do
  # (two work items, their factorisations in one set of launches like the timed region's; the timed region and the roofline pass each run them once)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu --no-config5 --no-config2 --no-config4 --no-host-api --no-plain-pass > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
unset HELM_ND_SPARSE_RHS
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc3d_$C -- python3 $GRAFT_REPO_ROOT/tools/apply3d_micro.py 16 4 > $OUT/pmc3d_$C.json 2> $OUT/pmc3d_$C.err
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_reduce.py --fetch $OUT/pmc_FETCH_SIZE --write $OUT/pmc_WRITE_SIZE --kernel k_zgemm3,k_gj_step --out $OUT/pmc_traffic_zgemm.json --note "all k_zgemm3 and k_gj_step dispatches of four work items (two in the timed region, two in the roofline pass; factorisations in sets of two + solve passes), every front computed (HELM_ND_SPARSE_RHS=0)" > /dev/null
python3 tools/pmc_reduce.py --fetch $OUT/pmc_FETCH_SIZE --write $OUT/pmc_WRITE_SIZE --kernel k_resid_nm --out $OUT/pmc_traffic_resid.json --note "node-major residual launches of four work items (direct output: the launch reads the caller's wavefield array and stores nothing), q read everywhere (HELM_ND_SPARSE_RHS=0)" > /dev/null
python3 tools/pmc_reduce.py --fetch $OUT/pmc_FETCH_SIZE --write $OUT/pmc_WRITE_SIZE --kernel k_stencil_t --out $OUT/pmc_traffic_stencil_micro.json --note "rhs-major stencil apply launches of the in-bench microbenchmark (B = 1, 8, 32, 64; 6 launches each)" > /dev/null
python3 tools/pmc_reduce.py --fetch $OUT/pmc3d_FETCH_SIZE --write $OUT/pmc3d_WRITE_SIZE --kernel k_stencil3 --grid 256 --batch 16 --out $OUT/pmc_traffic_stencil3.json --note "27-point apply, 256 x 256 x 128, B = 16, coefficients on the fly (tools/apply3d_micro.py 16 4: 4 launches); algorithmic bytes by SURVEY 8(d): N (32 B + 432) = 7.92 GB" > /dev/null
find $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc3d_FETCH_SIZE $OUT/pmc3d_WRITE_SIZE -name "*.csv" -size +2M -delete
python3 tools/apply3d_micro.py 16 > $OUT/apply3d_B16.json 2>/dev/null
python3 tools/factor_trace.py > /dev/null 2> $OUT/factor_trace_raw.txt
HELM_ND_TRACE=1 python3 tools/bench_direct.py --freqs 5.5 > $OUT/trace.txt 2>&1
HELM_GEMM_LOG=1 python3 tools/bench_direct.py --freqs 5.5 > /dev/null 2> $OUT/gemm_log_raw.txt
python3 tools/gemm_log.py $OUT/gemm_log_raw.txt 70 > $OUT/gemm_log.txt
rm -f $OUT/gemm_log_raw.txt
python3 tools/factor_many_probe.py 1024 5 > $OUT/factor_many.txt 2>/dev/null
python3 tools/c4_repeat.py 8 alternate > $OUT/c4_repeat.txt 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats3d -o s -- python3 $GRAFT_REPO_ROOT/tools/bench3d.py --freqs 5 --nsrc 16 > $OUT/bench3d_under_rocprof.txt 2> $OUT/stats3d.err
find $OUT/stats_pipe $OUT/stats_serial $OUT/stats_serial_sparse $OUT/stats_sets $OUT/stats3d -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*agent_info.csv" -delete
ls $OUT | head -60
