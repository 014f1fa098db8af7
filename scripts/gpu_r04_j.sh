#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed\|Error" $O/pytest_gpu.log | tail -5
echo "== mid-size (default routing)"; timeout 300 python scripts/midsize_profile.py > $O/midsize.json 2> $O/midsize.err; cat $O/midsize.json | cut -c1-200; tail -3 $O/midsize.err
echo "== mid-size (one launch everywhere, stamps)"; timeout 300 python scripts/midsize_profile.py mid_max_queries=131072 time_filter_launches=1 300 1024 4096 16384 65536 > $O/midsize1.json 2> $O/midsize1.err; cat $O/midsize1.json | cut -c1-300
echo "== reference grid"; timeout 600 python scripts/reference_grid_bench.py > $O/reference_grid.json 2> $O/reference_grid.err; tail -4 $O/reference_grid.err | cut -c1-300
echo "== small batches through the reference API"; timeout 300 python scripts/small_batch_latency.py --save > $O/small_batch.log 2>&1; tail -3 $O/small_batch.log | cut -c1-300
