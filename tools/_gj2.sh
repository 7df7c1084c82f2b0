#!/bin/bash
tools/_gj.sh 2>&1 | grep "k_gj32_inverse"
python3 tools/gj_lab.py 256 512 1024
python3 -m pytest tests/test_gpu_direct.py -x -q 2>&1 | tail -3
