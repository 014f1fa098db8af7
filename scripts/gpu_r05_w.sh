#!/bin/bash
# round 5: why the first mid-size measurement (300 proposals after the 262144-proposal warm-up) takes 200-300 us per call
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo "default"; timeout 300 python scripts/midsize_profile.py 300 300 1024 300 2>/dev/null | cut -c1-90
echo "fused_first_range=0"; timeout 300 python scripts/midsize_profile.py 300 300 1024 fused_first_range=0 2>/dev/null | cut -c1-90
echo "filter_second_range_pct=0"; timeout 300 python scripts/midsize_profile.py 300 300 filter_second_range_pct=0 2>/dev/null | cut -c1-90
