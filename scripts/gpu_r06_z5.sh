#!/bin/bash
# mid-size batches against filter_split_waves (waves a single-sweep launch aims at when it splits the tiles)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for W in 1024 2048 4096 8192 16384; do
echo "== filter_split_waves=$W"; timeout 200 python scripts/midsize_profile.py 16384 65536 131072 262144 filter_split_waves=$W 2>/dev/null | cut -c1-70
done
echo "== phased from 65536 on (filter_phase_min_queries=1024)"; timeout 200 python scripts/midsize_profile.py 65536 131072 262144 filter_phase_min_queries=1024 2>/dev/null | cut -c1-70
