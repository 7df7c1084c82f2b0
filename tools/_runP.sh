cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_p; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_direct.py -x -q -m gpu 2>&1 | tail -3
for x in 1 0; do
export HELM_ND_XCDMAP=$x
python3 bench.py --no-cpu --no-config5 --no-host-api --steps 16 --warmup 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xcdmap $x', round(d['value']), round(d['ms_per_step'],2), round(d['unprofiled']['value']), d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  HELM_ND_SPARSE_RHS=0 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${C}_$x -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-config5 --no-host-api --no-pipeline --no-plain-pass --no-roofline-pass > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_reduce.py --fetch $OUT/pmc_FETCH_SIZE_$x --write $OUT/pmc_WRITE_SIZE_$x --kernel k_zgemm3 --out $OUT/pmc_zgemm_$x.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('xcdmap $x GEMM traffic per item GB', d['traffic_bytes_per_launch']*d['launches_fetch_pass']/1e9, 'read', d['hbm_read_bytes_corrected']/1e9, 'write', d['hbm_write_bytes']/1e9)"
done
find $OUT -name "*.csv" -size +1M -delete
