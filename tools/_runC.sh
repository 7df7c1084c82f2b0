cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_c; mkdir -p $OUT
HELM_ND_GEMMV=8 timeout 600 python -m pytest tests/test_gpu_direct.py -x -q -m gpu -k "zgemm or inverse" 2>&1 | tail -5
timeout 900 python3 tools/zgemm_lab.py 7,8 > $OUT/lab.txt 2>&1
cat $OUT/lab.txt | grep -v "^\[{"
