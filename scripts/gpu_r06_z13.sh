#!/bin/bash
# experiment: k_sweep_min with s_setprio patterns (the younger workgroup of a CU at high priority for two of three tiles, the older
# for one) against the production build, same box, alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp ultranest_amd/libmlfriends_hip.so /tmp/prod.so
for round in 1 2 3; do
cp scripts/probes/bin/libmlfriends_prio.so ultranest_amd/libmlfriends_hip.so
echo "== prio"; timeout 200 python scripts/fused_ab.py 200 sq:4:3 2>/dev/null | tail -1 | cut -c1-170
cp /tmp/prod.so ultranest_amd/libmlfriends_hip.so
echo "== production"; timeout 200 python scripts/fused_ab.py 200 sq:4:3 2>/dev/null | tail -1 | cut -c1-170
done
