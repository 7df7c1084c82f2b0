#!/bin/bash
# config-5 set-up: products by shape, and a kernel trace of one set-up + solve
OUT=gpurun_out/c5a; mkdir -p $OUT
HELM_GEMM_LOG=1 python3 tools/bench3d.py --freqs 5 --nsrc 16 --no-apply > $OUT/bench3d.txt 2> $OUT/gemm_raw.txt
python3 tools/gemm_log.py $OUT/gemm_raw.txt 60 > $OUT/gemm_log.txt
rm -f $OUT/gemm_raw.txt
HELM_ND_TRACE=1 HELM_MG3_TRACE=1 python3 tools/bench3d.py --freqs 5 --nsrc 16 --no-apply > $OUT/trace.txt 2>&1
tail -40 $OUT/gemm_log.txt
