#!/bin/bash
# quick parameter sweeps of the 3-D multigrid preconditioner on config 5 (one frequency, 4 sources); usage: tools/sweep3d.sh [freq]
cd $GRAFT_REPO_ROOT
F=${1:-3}
run() { echo "== $*"; env "$@" python tools/bench3d.py --freqs $F --nsrc 4 --no-apply 2>&1 | grep "^solve" | cut -c1-160; }
run HELM_MG3_OMEGA=0.8
run HELM_MG3_OMEGA=0.9
run HELM_MG3_OMEGA=1.0
run HELM_MG3_OMEGA=1.1
run HELM_MG3_OMEGA=0.9 HELM_MG3_BETA=12.0
run HELM_MG3_OMEGA=1.0 HELM_MG3_BETA=12.0
run HELM_MG3_OMEGA=1.0 HELM_MG3_NU1=2 HELM_MG3_NU2=1
