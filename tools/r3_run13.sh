cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3o
timeout 900 python -m pytest tests/test_gpu_direct.py tests/test_gpu_layouts.py -x -q -m gpu > gpurun_out/r3o/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3o/tests.log
tail -4 gpurun_out/r3o/tests.log
for L in 1 0 1 0; do
HELM_ND_XCDMAP=$L timeout 600 python bench.py --steps 32 --warmup 5 --no-cpu --no-config5 --no-host-api 2> /dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('xcdmap $L', d['value'], d['ms_per_step'], d['unprofiled']['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
"
done
for L in 1 0; do HELM_ND_XCDMAP=$L HELM_ND_TRACE=1 python3 tools/bench_direct.py --freqs 5.5 2>&1 | grep "nd trace" | grep -E "total" | head -3 | tr '\n' ' '; echo; done
