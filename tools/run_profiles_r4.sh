#!/bin/bash
# Round-4 profile collection on the GPU box (run from the repo root through gpurun):
#   tools/run_profiles_r4.sh <outdir under gpurun_out> [githash]
# 1. python bench.py (all legs)                          -> bench_n1.json      2. the driver's command line (--steps 20 --warmup 5) -> bench_driver.json
# 3. rocprofv3 --kernel-trace --stats of the pipelined region and of the serial kernel pass (separate runs)
# 4. two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of one serial work item, every front computed -> HBM bytes per GEMM / residual launch
# 5. per-level trace, per-launch GEMM log, fp64 probe with SQ counters, config 5 under rocprofv3
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
echo "${2:-unknown}" > $OUT/githash.txt
python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python3 bench.py --steps 20 --warmup 5 --no-cpu > $OUT/bench_driver.json 2> $OUT/bench_driver.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pipe -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-config5 --no-host-api --steps 8 --warmup 2 --no-plain-pass > $OUT/bench_pipelined_under_rocprof.json 2> $OUT/stats_pipe.err
HELM_ND_SPARSE_RHS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-config5 --no-host-api --no-pipeline --steps 8 --warmup 2 --no-plain-pass > $OUT/bench_serial_under_rocprof.json 2> $OUT/stats_serial.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial_sparse -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-config5 --no-host-api --no-pipeline --steps 8 --warmup 2 --no-plain-pass --no-roofline-pass > $OUT/bench_serial_sparse_under_rocprof.json 2> $OUT/stats_serial_sparse.err
export HELM_ND_SPARSE_RHS=0
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-config5 --no-host-api --no-pipeline --no-plain-pass > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
unset HELM_ND_SPARSE_RHS
cd $GRAFT_REPO_ROOT
python3 tools/pmc_reduce.py --fetch $OUT/pmc_FETCH_SIZE --write $OUT/pmc_WRITE_SIZE --kernel k_zgemm3,k_gj_step --out $OUT/pmc_traffic_zgemm.json --note "all k_zgemm3 and k_gj_step (one-launch Gauss-Jordan block step, booked with the products) dispatches of one work item (factorisation + solve passes), serial, every front computed (HELM_ND_SPARSE_RHS=0)" > /dev/null
python3 tools/pmc_reduce.py --fetch $OUT/pmc_FETCH_SIZE --write $OUT/pmc_WRITE_SIZE --kernel k_resid_nm --out $OUT/pmc_traffic_resid.json --note "node-major residual launches (9-point stencil apply + q operand + wavefield store) of one work item, q read everywhere (HELM_ND_SPARSE_RHS=0)" > /dev/null
python3 tools/pmc_reduce.py --fetch $OUT/pmc_FETCH_SIZE --write $OUT/pmc_WRITE_SIZE --kernel k_stencil_t --out $OUT/pmc_traffic_stencil_micro.json --note "rhs-major stencil apply launches of the in-bench microbenchmark (B = 1, 8, 32, 64; 6 launches each)" > /dev/null
find $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE -name "*.csv" -size +2M -delete
HELM_ND_TRACE=1 python3 tools/bench_direct.py --freqs 5.5 > $OUT/trace.txt 2>&1
HELM_ND_SPARSE_RHS=0 HELM_ND_TRACE=1 python3 tools/bench_direct.py --freqs 5.5 > $OUT/trace_every_front.txt 2>&1
HELM_GEMM_LOG=1 python3 tools/bench_direct.py --freqs 5.5 > /dev/null 2> $OUT/gemm_log_raw.txt
python3 tools/gemm_log.py $OUT/gemm_log_raw.txt 70 > $OUT/gemm_log.txt
rm -f $OUT/gemm_log_raw.txt
timeout 300 tools/fp64_clock 1.0 > $OUT/fp64_clock.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_probe -- $GRAFT_REPO_ROOT/tools/fp64_clock 0.1 > $OUT/fp64_clock_pmc.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_gemm_sq -- python3 $GRAFT_REPO_ROOT/tools/zgemm_lab.py 7 "s256 Schur" 3 > $OUT/pmc_gemm_sq.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_gemm_sq2 -- python3 $GRAFT_REPO_ROOT/tools/zgemm_lab.py 7 "s256 Schur" 3 > $OUT/pmc_gemm_sq2.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats3d -o s -- python3 $GRAFT_REPO_ROOT/tools/bench3d.py --freqs 5 --nsrc 16 > $OUT/bench3d_under_rocprof.txt 2> $OUT/stats3d.err
find $OUT/stats_pipe $OUT/stats_serial $OUT/stats_serial_sparse $OUT/stats3d $OUT/pmc_probe $OUT/pmc_gemm_sq $OUT/pmc_gemm_sq2 -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*agent_info.csv" -delete
cd $GRAFT_REPO_ROOT
python3 tools/zgemm_lab.py 1,7 > $OUT/zgemm_lab.txt 2>&1
ls -la $OUT
