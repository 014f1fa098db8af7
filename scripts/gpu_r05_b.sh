#!/bin/bash
# round 5, call b: the min-only form of the one-launch path -- parity, then the mid-size table for the three routings
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== gpu tests"; timeout 1200 python -m pytest tests -m gpu -x -q > $O/r05b_pytest.log 2>&1; grep -E "passed|failed|error" $O/r05b_pytest.log | tail -3; grep -B5 -A25 "^E " $O/r05b_pytest.log | head -60
for setting in "mid_max_queries=300000 mid_min=1" "mid_max_queries=300000 mid_min=0" "mid_max_queries=0"; do
  echo "== midsize: $setting"
  timeout 300 python scripts/midsize_profile.py $setting 2> $O/r05b_mid.err | tee -a $O/r05b_midsize.jsonl | cut -c1-200
  tail -2 $O/r05b_mid.err | grep -v amdgpu.ids
done
echo "== stamps (time_filter_launches=1) at 300 / 16384 / 65536, min-only form"
timeout 300 python scripts/midsize_profile.py 300 16384 65536 mid_max_queries=300000 mid_min=1 time_filter_launches=1 2>/dev/null | tee $O/r05b_mid_stamps.jsonl | cut -c1-400
