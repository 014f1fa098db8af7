#!/bin/bash
# round 5, call e: dimensionalities above 128 (mlf_wide.hip) -- parity, the d = 100 / d = 256 configurations, the reference's own grid
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -x -q > $O/r05e_pytest.log 2>&1; grep -E "passed|failed|error" $O/r05e_pytest.log | tail -3; grep -B5 -A30 "^E " $O/r05e_pytest.log | head -90
echo "== config bench (d = 100, d = 256)"; timeout 900 python scripts/config_bench.py published-shape C5-at-d100 wide > $O/r05e_config_bench.json 2> $O/r05e_config_bench.err; tail -3 $O/r05e_config_bench.err | cut -c1-300; head -c 3000 $O/r05e_config_bench.json
echo "== reference grid"; timeout 900 python scripts/reference_grid_bench.py > $O/r05e_reference_grid.json 2> $O/r05e_reference_grid.err; grep "transform" $O/r05e_reference_grid.err | tail -8 | cut -c1-330
