cd $GRAFT_REPO_ROOT
for cfg in "300 768" "600 1024" "600 1536" "1200 2048" "400 1024"; do
set -- $cfg
HELM_ND_SPLITK_TILES=$1 HELM_ND_SPLITK_WGS=$2 HELM_ND_TRACE=1 timeout 600 python tools/bench3d.py --freqs 5 --nsrc 16 2>&1 | grep "nd trace" | grep "forward  total\|backward total" | sed -n 3,4p | tr '\n' ' '; echo " <- tiles<$1 wgs $2"
done
