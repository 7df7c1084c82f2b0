#!/bin/bash
python3 tools/tile_lab.py | cut -c1-34
python3 bench.py --steps 20 --warmup 5 --no-config5 > gpurun_out/b_tl.json 2> gpurun_out/b_tl.err
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/b_tl.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'], d['config']['driver_visible'])
PY
