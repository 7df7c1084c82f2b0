cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_i; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_direct.py -x -q -m gpu 2>&1 | tail -4
for d in 0 1 2 4 7; do
echo "dbg $d: $(HELM_LEAF_DBG=$d HELM_ND_TRACE=1 python3 tools/bench_direct.py --freqs 5.5 2>&1 | grep 'nd trace' | head -2 | tr '\n' ' ')"
done
