cd $GRAFT_REPO_ROOT
for v in 1 0; do
echo "SPARSE_LEAF=$v: $(HELM_ND_SPARSE_LEAF=$v HELM_ND_TRACE=1 python3 tools/bench_direct.py --freqs 5.5 2>&1 | grep 'nd trace' | sed -n 18,22p | awk '{print $3,$4,$5,$(NF-1)}' | tr '\n' ';')"
done
echo "SPARSE_RHS=0: $(HELM_ND_SPARSE_RHS=0 HELM_ND_TRACE=1 python3 tools/bench_direct.py --freqs 5.5 2>&1 | grep 'nd trace' | sed -n 18,22p | awk '{print $3,$4,$5,$(NF-1)}' | tr '\n' ';')"
