cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/suite/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/suite/tests.log
tail -15 gpurun_out/suite/tests.log
