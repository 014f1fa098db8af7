#!/bin/bash
# the one-launch path (k_inside_mid) beyond its default limit of 2048 proposals
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for M in 2048 4096 8192 16384 32768; do
echo "== mid_max_queries=$M"; timeout 200 python scripts/midsize_profile.py 1024 2048 4096 8192 16384 32768 mid_max_queries=$M 2>/dev/null | cut -c1-64
done
