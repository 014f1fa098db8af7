#!/bin/bash
# same box, three builds: HEAD's library (scripts/probes/bin/libmlfriends_prev.so), the tree's, and the tree's with three of the stage
# stamps moved INSIDE the group (libmlfriends_stamps.so: 4 = after the chains of group 0, 6 = after the decisions of group 1,
# 8 = after the packing of group 2)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
cp ultranest_amd/libmlfriends_hip.so /tmp/new.so
for round in 1 2; do
echo "== new build, set E"; timeout 200 python scripts/fused_ab.py 200 w4:4:1 2>/dev/null | head -3 | cut -c1-200
cp scripts/probes/bin/libmlfriends_prev.so ultranest_amd/libmlfriends_hip.so
echo "== previous build, set E"; timeout 200 python scripts/fused_ab.py 200 w4:4:1 2>/dev/null | head -3 | cut -c1-200
cp /tmp/new.so ultranest_amd/libmlfriends_hip.so
done
cp scripts/probes/bin/libmlfriends_stamps.so ultranest_amd/libmlfriends_hip.so
echo "== stamps build"; timeout 200 python scripts/fused_ab.py 200 w4:4:1 2>/dev/null | tail -1 > $O/r06_stamps_inside.json; cut -c1-1500 $O/r06_stamps_inside.json
cp /tmp/new.so ultranest_amd/libmlfriends_hip.so
echo "== mid-size, new build"; timeout 200 python scripts/midsize_profile.py 16384 65536 131072 2>/dev/null | cut -c1-100
echo "== parity"; timeout 900 python -m pytest tests/test_config_sizes.py tests/test_gpu_filter.py tests/test_prep4_bounds.py -m gpu -x -q 2>&1 | tail -2
