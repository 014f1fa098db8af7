#!/bin/bash
# same box: the tree's library (no s_setprio for the ring walker's wave) against HEAD's (with it)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp ultranest_amd/libmlfriends_hip.so /tmp/new.so
for round in 1 2 3; do
echo "== without setprio"; timeout 200 python scripts/walk_rounds_profile.py 1500 2>/dev/null | cut -c1-230
cp scripts/probes/bin/libmlfriends_prev.so ultranest_amd/libmlfriends_hip.so
echo "== with setprio"; timeout 200 python scripts/walk_rounds_profile.py 1500 2>/dev/null | cut -c1-230
cp /tmp/new.so ultranest_amd/libmlfriends_hip.so
done
