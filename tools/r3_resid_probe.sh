cd $GRAFT_REPO_ROOT
HELM_ND_DEBUG=1 python3 tools/bench_direct.py --grid 512 --batch 64 --freqs 6,7,8,9,10,11,12,13,14,15,16,17,18,19 2>&1 | grep -E "pass 1|freq" | awk '/pass 1/{c++; if (c%3==1) print} /"freq"/{print substr($0,1,40)}' | head -60
