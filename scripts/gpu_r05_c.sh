#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== stamps of the last arrival of set 0 (indices 2..6), min-only form"
timeout 300 python scripts/midsize_profile.py 300 4096 16384 65536 131072 mid_max_queries=300000 mid_min=1 time_filter_launches=1 2>/dev/null | cut -c1-300
echo "== kernel trace at 65536 / 16384, both forms"
export TMPDIR=/tmp
for m in 1 0; do
  rm -rf /tmp/prof_mid; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_mid -o mid -- python $R/scripts/midsize_profile.py 16384 65536 mid_max_queries=300000 mid_min=$m > /dev/null 2>&1)
  f=$(find /tmp/prof_mid -name "*kernel_stats.csv" | head -1); echo "mid_min=$m: $f"; head -8 "$f" | cut -c1-200
done
echo "== hbm probe"; ./scripts/probes/bin/hbm_stream_probe | tee $O/r05_hbm_stream_probe.json | cut -c1-200
