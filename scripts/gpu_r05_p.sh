#!/bin/bash
# round 5: three ranges + per-proposal stage inside the first launch as the default -- filter / full-size tests, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_filter.py tests/test_config_sizes.py -m gpu -x -q --durations=5 > $O/r05p_pytest.log 2>&1; grep -E "passed|failed|error" $O/r05p_pytest.log | tail -3; grep -B5 -A30 "^E " $O/r05p_pytest.log | head -80
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r05p_bench.json 2> $O/r05p_bench.err; tail -3 $O/r05p_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05p_bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['ms_per_launch_by_phase'], r['executed_mfma_per_launch_by_phase'], r['range_cuts_tiles'])
print(json.dumps(r['first_launch'])[:600]); print(d['kernel_ms']['prep'], d['kernel_ms']['scan'], d['kernel_ms']['tail'], d['kernel_ms']['unfused'])
print(d['roofline_prep'] and d['roofline_prep']['achieved'], d['rebuild_ms'], d['rebuild_device_resident_ms'], d['cpu_baseline'] and d['cpu_baseline']['value'])
PY
