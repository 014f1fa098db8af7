#!/bin/bash
# the full-size tests with the batches resident in HBM (phased routing, same-form variant)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_config_sizes.py -m gpu -x -q -k "same_quadratic or full_size" > $O/r06y2_tests.log 2>&1; tail -3 $O/r06y2_tests.log; grep -B5 -A25 "^E " $O/r06y2_tests.log | head -60
