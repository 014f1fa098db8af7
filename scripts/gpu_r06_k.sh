#!/bin/bash
# same box, two builds: the library of the previous commit (scripts/probes/bin/libmlfriends_prev.so, built locally) against the tree's
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
cp ultranest_amd/libmlfriends_hip.so /tmp/new.so
for round in 1 2; do
echo "== new build, set E"; timeout 200 python scripts/fused_ab.py 200 w4:4:1 2>/dev/null | head -3 | cut -c1-200
echo "== new build, set U"; MLF_AB_SET=U timeout 200 python scripts/fused_ab.py 200 w4:4:1 2>/dev/null | head -3 | cut -c1-200
cp scripts/probes/bin/libmlfriends_prev.so ultranest_amd/libmlfriends_hip.so
echo "== previous build, set E"; timeout 200 python scripts/fused_ab.py 200 w4:4:1 2>/dev/null | head -3 | cut -c1-200
echo "== previous build, set U"; MLF_AB_SET=U timeout 200 python scripts/fused_ab.py 200 w4:4:1 2>/dev/null | head -3 | cut -c1-200
cp /tmp/new.so ultranest_amd/libmlfriends_hip.so
done
echo "== mid-size, new build"; timeout 200 python scripts/midsize_profile.py 16384 65536 131072 2>/dev/null | cut -c1-100
cp scripts/probes/bin/libmlfriends_prev.so ultranest_amd/libmlfriends_hip.so
echo "== mid-size, previous build"; timeout 200 python scripts/midsize_profile.py 16384 65536 131072 2>/dev/null | cut -c1-100
cp /tmp/new.so ultranest_amd/libmlfriends_hip.so
echo "== parity"; timeout 600 python -m pytest tests/test_config_sizes.py tests/test_gpu_filter.py -m gpu -x -q 2>&1 | tail -2
