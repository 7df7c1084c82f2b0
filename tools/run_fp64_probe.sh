#!/bin/bash
# fp64 ceiling probe on the GPU box (through gpurun): tools/run_fp64_probe.sh <outdir under gpurun_out>
# 1. tools/fp64_clock at 1, 2, 4, 8 waves per SIMD (occupancy pinned by LDS)   2. the same under rocprofv3 --pmc (SQ counters), short runs
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
[ -x tools/fp64_clock ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/fp64_clock.hip -o tools/fp64_clock
timeout 300 tools/fp64_clock 1.0 > $OUT/fp64_clock.txt 2>&1
(rocm-smi --showclocks --showpower 2>&1 | tail -15) >> $OUT/fp64_clock.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_probe -- $GRAFT_REPO_ROOT/tools/fp64_clock 0.1 > $OUT/fp64_clock_pmc.txt 2>&1
ls -R $OUT | head -30
