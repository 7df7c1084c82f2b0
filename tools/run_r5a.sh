#!/bin/bash
# round 5, first GPU call: the GPU test suite on the split / pruned library, the bench line, and a kernel trace (start / end of every launch) of the
# pipelined timed region, from which tools/trace_overlap.py computes how much of the wall time the chip runs one, two or no kernels
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
timeout 600 python3 bench.py --no-cpu > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_pipe -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-config5 --no-host-api --steps 8 --warmup 3 --no-plain-pass --no-roofline-pass > $OUT/bench_traced.json 2> $OUT/trace_pipe.err
find $OUT -name "*agent_info.csv" -delete
ls -la $OUT $OUT/trace_pipe/* | head -30
