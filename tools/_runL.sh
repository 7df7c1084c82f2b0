cd $GRAFT_REPO_ROOT
for r in 3000 1000 500 250; do
echo "RECURSE_N $r: $(HELM_ND_RECURSE_N=$r HELM_ND_TRACE=1 python3 tools/bench_direct.py --freqs 5.5 2>&1 | grep 'nd trace' | head -17 | grep -E 'level  [0-4] |total' | awk '{print $(NF-1)}' | tr '\n' ' ')"
done
for l in 0 1; do
echo "LOOKAHEAD $l: $(HELM_ND_LOOKAHEAD=$l HELM_ND_TRACE=1 python3 tools/bench_direct.py --freqs 5.5 2>&1 | grep 'nd trace' | head -17 | grep -E 'level  [0-4] |total' | awk '{print $(NF-1)}' | tr '\n' ' ')"
done
