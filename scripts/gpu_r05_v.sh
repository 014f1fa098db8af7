#!/bin/bash
# round 5: k_uncertain without the fill kernel in front of it -- filter / full-size tests, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_filter.py tests/test_config_sizes.py tests/test_small_path.py -m gpu -x -q > $O/r05v_pytest.log 2>&1; grep -E "passed|failed|error" $O/r05v_pytest.log | tail -3; grep -B5 -A30 "^E " $O/r05v_pytest.log | head -80
timeout 600 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu > $O/r05v_bench.json 2> $O/r05v_bench.err; tail -3 $O/r05v_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05v_bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['ms_per_launch_by_phase'], d['batch_counters'])
PY
timeout 300 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
