#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== walker tests"
timeout 900 python -m pytest tests/test_popstepsampler.py tests/test_harness.py tests/test_stepfuncs_golden.py -m gpu -x -q > $O/r06e_tests.log 2>&1; tail -15 $O/r06e_tests.log | cut -c1-300
echo "== e2e (population slice sampler, multi-round device loop, state in registers)"
timeout 600 python scripts/e2e_run.py nsteps10=40,80 > $O/r06e_e2e.log 2>&1; cut -c1-330 $O/r06e_e2e.log | tail -5
cp $O/e2e_run.json $O/r06e_e2e_run.json 2>/dev/null
echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -x -q > $O/r06e_pytest_gpu.log 2>&1; grep -n "passed\|failed\|Error" $O/r06e_pytest_gpu.log | tail -5; grep -B5 -A25 "^E " $O/r06e_pytest_gpu.log | head -60
