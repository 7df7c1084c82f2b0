#!/bin/bash
python3 -m pytest tests/test_gpu_direct.py tests/test_gpu_stable_fronts.py -x -q 2>&1 | tail -3
python3 bench.py --steps 20 --warmup 5 --no-config5 > gpurun_out/b_x2.json 2> gpurun_out/b_x2.err
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/b_x2.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'], d['config']['driver_visible'])
PY
