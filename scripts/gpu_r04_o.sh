#!/bin/bash
# k_inside_mid with the re-check's operands brought forward (layer matrix in LDS, live rows requested before the whitening)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== filter / size / region tests"; timeout 900 python -m pytest tests/test_gpu_filter.py tests/test_config_sizes.py tests/test_regions.py tests/test_small_path.py -m gpu -x -q > $O/pytest_o.log 2>&1; grep -n "passed\|failed\|Error" $O/pytest_o.log | tail -5; tail -30 $O/pytest_o.log | grep -v "^$" | cut -c1-200 | head -30
echo "== mid-size (default routing)"; timeout 300 python scripts/midsize_profile.py > $O/midsize_o.json 2> $O/midsize_o.err; cat $O/midsize_o.json | cut -c1-200; tail -3 $O/midsize_o.err
echo "== mid-size (one launch everywhere, stamps)"; timeout 300 python scripts/midsize_profile.py mid_max_queries=131072 time_filter_launches=1 300 1024 2048 4096 16384 > $O/midsize_o1.json 2> $O/midsize_o1.err; cat $O/midsize_o1.json | cut -c1-300
