#!/bin/bash
# k_prep4<.., SQ>: parity (filter tests with the per-proposal stage in its own launch, mid-size routing, sizes), mid-size timings against the build before
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_filter.py tests/test_config_sizes.py tests/test_philox.py tests/test_regions.py tests/test_small_path.py -m gpu -x -q > $O/r06z7_tests.log 2>&1; tail -3 $O/r06z7_tests.log; grep -B5 -A25 "^E " $O/r06z7_tests.log | head -60
cp ultranest_amd/libmlfriends_hip.so /tmp/new.so
for round in 1 2; do
echo "== new"; timeout 200 python scripts/midsize_profile.py 4096 16384 65536 131072 2>/dev/null | cut -c1-60
cp scripts/probes/bin/libmlfriends_prev.so ultranest_amd/libmlfriends_hip.so
echo "== before"; timeout 200 python scripts/midsize_profile.py 4096 16384 65536 131072 2>/dev/null | cut -c1-60
cp /tmp/new.so ultranest_amd/libmlfriends_hip.so
done
