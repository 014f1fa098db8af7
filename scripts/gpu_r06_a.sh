#!/bin/bash
# round 6, call a: new tests first (rounds == single steps, per-device grants, oracle slice, netiter on the box), the e2e runs with
# the multi-round walker loop, then the whole GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== new tests"
timeout 900 python -m pytest tests/test_popstepsampler.py tests/test_netiter.py "tests/test_gpu_filter.py::test_lds_grants_are_per_device_and_reissued" "tests/test_gpu_filter.py::test_options_are_per_region_handle" "tests/test_config_sizes.py::test_c5_full_size_filter_equals_exact_scan" -m gpu -x -q > $O/r06a_new_tests.log 2>&1; tail -15 $O/r06a_new_tests.log | cut -c1-300
echo "== e2e"
timeout 900 python scripts/e2e_run.py 400000 > $O/r06a_e2e.log 2>&1; tail -3 $O/r06a_e2e.log | cut -c1-900
cp $O/e2e_run.json $O/r06a_e2e_run.json 2>/dev/null
echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -x -q > $O/r06a_pytest_gpu.log 2>&1; grep -n "passed\|failed\|Error" $O/r06a_pytest_gpu.log | tail -5; grep -B5 -A25 "^E " $O/r06a_pytest_gpu.log | head -60
