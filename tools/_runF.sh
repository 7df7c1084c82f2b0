cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_f; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_dispatch.py tests/test_gpu_layouts.py tests/test_gpu_3d.py tests/test_gpu_solver.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8 > $OUT/pytest_gpu.txt
cat $OUT/pytest_gpu.txt
