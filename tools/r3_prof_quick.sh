set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-config5 --no-host-api --no-pipeline --no-plain-pass --steps 8 --warmup 2 > $OUT/bench_serial_under_rocprof.json 2> $OUT/stats.err
find $OUT/stats -name "*kernel_trace.csv" -size +8M -delete
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,os,re
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r3f/stats/**/s_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(int(r['TotalDurationNs']) for r in rows)
for r in rows[:28]:
    n=r['Name'].replace('(anonymous namespace)::','').replace('HIP_vector_type<double, 2u>','cplx')
    print('%-70s %7s calls %9.1f us avg %6.2f %%' % (n[:70], r['Calls'], float(r['AverageNs'])/1e3, 100.0*int(r['TotalDurationNs'])/tot))
print('total ms per step', tot/1e6/(8+2+8))
PY
