cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_g; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_stable_fronts.py tests/test_25d_pairing.py -x -q -m gpu 2>&1 | tail -15 > $OUT/pytest_gpu.txt
cat $OUT/pytest_gpu.txt
