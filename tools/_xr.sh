#!/bin/bash
for V in 2 1 2 1; do
HELM_ND_XR_NT=$V python3 bench.py --steps 20 --warmup 5 --no-config5 --no-cpu --no-host-api > gpurun_out/b_gs.json 2> gpurun_out/b_gs.err
python3 - "$V" <<'PY'
import json, sys
d = json.load(open('gpurun_out/b_gs.json'))
v = d['config']['driver_visible']; ip = d['roofline']['in_pipeline']
print('xr_nt', sys.argv[1], 'value %.0f ms %.2f unprof %.0f dense %.0f strong %.4f frac %.4f inpipe %.4f' % (d['value'], d['ms_per_step'], v['unprofiled_value'], v['every_front_computed_value'], v['strong_job_seconds'], v['roofline_frac_serial'], v['roofline_in_pipeline_frac']))
PY
done
