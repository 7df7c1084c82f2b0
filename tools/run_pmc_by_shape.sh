#!/bin/bash
# traffic of the products by shape: tools/_pmc.sh <HELM_ND_XCDMAP> <tag>
export HELM_ND_XCDMAP=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcd_$2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE
export HELM_ND_SPARSE_RHS=0
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$C -- python3 $GRAFT_REPO_ROOT/tools/bench_direct.py --freqs 5.5 > $OUT/run_$C.txt 2>&1
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_by_dispatch.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE k_zgemm3,k_gj_step $OUT/seq.txt > $OUT/by_dispatch.txt
HELM_GEMM_LOG=1 python3 tools/bench_direct.py --freqs 5.5 > $OUT/bench_direct.txt 2> /tmp/gl.txt
python3 tools/gemm_log.py /tmp/gl.txt 90 > $OUT/gemm_log.txt
grep 'gemm log' /tmp/gl.txt > $OUT/gemm_seq.txt
python3 tools/pmc_pair.py $OUT/seq.txt $OUT/gemm_seq.txt 45 > $OUT/pairs.txt
head -1 $OUT/pairs.txt; head -1 $OUT/gemm_log.txt; grep -v amdgpu $OUT/bench_direct.txt | tail -3
