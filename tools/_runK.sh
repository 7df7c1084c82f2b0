cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_k; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-config5 --no-host-api --no-pipeline --steps 8 --warmup 2 --no-plain-pass > $OUT/bench_serial_under_rocprof.json 2> $OUT/stats_serial.err
find $OUT/stats_serial -name "*kernel_trace.csv" -size +4M -delete
