#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_philox.py -m gpu -x -q > $O/r06z3_tests.log 2>&1; tail -3 $O/r06z3_tests.log; grep -B10 -A25 "^E " $O/r06z3_tests.log | head -80
