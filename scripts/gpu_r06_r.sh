#!/bin/bash
# experiment: k_sweep_min with the workgroups of every second block of 256 started late (MLF_SWEEP_DELAY x 64 s_sleep units =
# x 4096 cycles): do the lockstep rounds cost what DESIGN 9 says?
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp ultranest_amd/libmlfriends_hip.so /tmp/new.so
cp scripts/probes/bin/libmlfriends_delay.so ultranest_amd/libmlfriends_hip.so
for D in 0 1 2 3 4 6 8 0; do
echo "== delay $D"; MLF_SWEEP_DELAY=$D timeout 200 python scripts/fused_ab.py 150 w4:4:1 2>/dev/null | head -2 | tail -1 | cut -c1-180
done
cp /tmp/new.so ultranest_amd/libmlfriends_hip.so
