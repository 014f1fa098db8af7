#!/bin/bash
# k_uncertain: the two-stage tile body against HEAD's (scripts/probes/bin/libmlfriends_prev.so) on one box; probe B = per-wave
# duration of the sweep stage in workgroup 0 with the new body (probe A / A2 of the old body: 30-40 000 / 23-37 000 cycles)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp ultranest_amd/libmlfriends_hip.so /tmp/new.so
cp scripts/probes/bin/libmlfriends_probeB.so ultranest_amd/libmlfriends_hip.so
echo "== probe B"; timeout 200 python scripts/uncertain_probe.py 2>/dev/null | tail -2
for round in 1 2; do
cp /tmp/new.so ultranest_amd/libmlfriends_hip.so
echo "== new build"; timeout 200 python scripts/fused_ab.py 200 w4:4:1 2>/dev/null | head -2 | cut -c1-200
timeout 100 python scripts/uncertain_probe.py 2>/dev/null | tail -1
cp scripts/probes/bin/libmlfriends_prev.so ultranest_amd/libmlfriends_hip.so
echo "== previous build"; timeout 200 python scripts/fused_ab.py 200 w4:4:1 2>/dev/null | head -2 | cut -c1-200
timeout 100 python scripts/uncertain_probe.py 2>/dev/null | tail -1
done
cp /tmp/new.so ultranest_amd/libmlfriends_hip.so
echo "== mid-size"; timeout 200 python scripts/midsize_profile.py 262144 2>/dev/null | cut -c1-300
echo "== parity"; timeout 900 python -m pytest tests/test_config_sizes.py tests/test_gpu_filter.py -m gpu -x -q 2>&1 | tail -2
