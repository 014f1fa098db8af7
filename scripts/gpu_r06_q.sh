#!/bin/bash
# two builds on one box: HEAD's library (scripts/probes/bin/libmlfriends_prev.so) against the tree's
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp ultranest_amd/libmlfriends_hip.so /tmp/new.so
for round in 1 2 3; do
echo "== new build"; timeout 200 python scripts/fused_ab.py 200 w4:4:1 2>/dev/null | head -2 | tail -1 | cut -c1-200
cp scripts/probes/bin/libmlfriends_prev.so ultranest_amd/libmlfriends_hip.so
echo "== previous build"; timeout 200 python scripts/fused_ab.py 200 w4:4:1 2>/dev/null | head -2 | tail -1 | cut -c1-200
cp /tmp/new.so ultranest_amd/libmlfriends_hip.so
done
echo "== parity"; timeout 900 python -m pytest tests/test_config_sizes.py tests/test_gpu_filter.py -m gpu -x -q 2>&1 | tail -2
