cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_e2; mkdir -p $OUT
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $OUT/pytest_gpu.txt
cat $OUT/pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
