#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for setting in "mid_max_queries=2048 mid_min=0" "mid_max_queries=2048 mid_min=0 filter_split_waves=4096" "mid_max_queries=2048 mid_min=0 filter_split_waves=8192" "mid_max_queries=2048 mid_min=0 filter_split_waves=1024" "mid_max_queries=2048 mid_min=0 filter_phase_min_queries=32767"; do
echo "== $setting"
timeout 300 python scripts/midsize_profile.py 4096 16384 32768 65536 131072 200000 262144 $setting 2>/dev/null | cut -c1-120
done
