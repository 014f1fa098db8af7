#!/bin/bash
# probe C: s_memtime against s_memrealtime (100 MHz) at the first four stage boundaries of k_uncertain's workgroup 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp ultranest_amd/libmlfriends_hip.so /tmp/new.so
cp scripts/probes/bin/libmlfriends_probeC.so ultranest_amd/libmlfriends_hip.so
echo "== probe C"; timeout 200 python scripts/uncertain_probe.py 2>/dev/null | tail -3
cp /tmp/new.so ultranest_amd/libmlfriends_hip.so
