#!/bin/bash
# round 6, call c: allocation trace of the timed region; host profile of config 4; build_front A/B under the profiler
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
tools/run_repro_r6.sh $1 1
sed -n '/timed region starts/,/timed region ends/p' $OUT/run1.err | grep -v "first launch" | head -40
python3 tools/profile_c4.py > $OUT/c4_profile.txt 2>&1
tail -90 $OUT/c4_profile.txt
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --no-cpu --no-config5 --no-config2 --no-config4 --no-host-api --steps 8 --warmup 3 --no-plain-pass --no-pipeline --no-roofline-pass"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bf4 -o s -- python3 $B > $OUT/bf4.json 2> $OUT/bf4.err
HELM_ND_BUILD1=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bf1 -o s -- python3 $B > $OUT/bf1.json 2> $OUT/bf1.err
for d in bf4 bf1; do echo $d; grep -h "k_nd_build_front\|k_fwd_flags" $(find $OUT/$d -name "*kernel_stats.csv") | cut -c1-200; done
find $OUT -name "*kernel_trace.csv" -size +4M -delete; find $OUT -name "*agent_info.csv" -delete
