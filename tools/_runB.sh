cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_b; mkdir -p $OUT
HELM_GEMM_LOG=1 python3 tools/bench_direct.py --freqs 5.5 > $OUT/bd.json 2> $OUT/gemm_log.txt
python3 tools/gemm_log.py $OUT/gemm_log.txt 60
