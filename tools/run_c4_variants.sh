#!/bin/bash
# config-4 stall hunt: dpred / Jtvec seconds of tools/profile_c4.py under variants of the stream set-up (three processes each)
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; for i in 1 2 3; do env $1 python3 tools/profile_c4.py 2>&1 | grep "===" | tr '\n' ' '; echo; done; }
run "X=1"
run "HELM_PF_PRIO=0"
run "HELM_PF_PRIO=-1"
run "HELM_ND_STABLE=0"
run "GPU_MAX_HW_QUEUES=1"
run "GPU_MAX_HW_QUEUES=8"
run "HIP_FORCE_DEV_KERNARG=0"
