#!/bin/bash
# default rebuild with mlf_cluster_labels in update_clusters and the folded range checks: parity tests, call breakdown, modes
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== region / layer / rebuild tests"; timeout 900 python -m pytest tests/test_regions.py tests/test_device_rebuild.py tests/test_reference_fixtures.py tests/test_harness.py tests/test_distributed.py -m gpu -x -q > $O/pytest_m.log 2>&1; grep -n "passed\|failed\|Error" $O/pytest_m.log | tail -5; tail -30 $O/pytest_m.log | grep -v "^$" | cut -c1-200 | head -30
echo "== rebuild calls"; timeout 300 python scripts/rebuild_calls.py > $O/rebuild_calls.log 2>&1; head -32 $O/rebuild_calls.log | cut -c1-150
echo "== rebuild modes"; timeout 300 python scripts/rebuild_modes.py > $O/rebuild_modes.json 2> $O/rebuild_modes.err; cut -c1-600 $O/rebuild_modes.json; tail -2 $O/rebuild_modes.err
