cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/p3d
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats3d -o s -- python3 $GRAFT_REPO_ROOT/tools/bench3d.py --freqs 4 --nsrc 16 > $OUT/bench3d.txt 2> $OUT/stats3d.err
find $OUT/stats3d -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -c 600 $OUT/bench3d.txt
