#!/bin/bash
# round 5: min-only sweep in three ranges -- parity tests, then the A/B against two ranges
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_filter.py tests/test_config_sizes.py -m gpu -x -q -k "three or optional_routings" > $O/r05l_pytest.log 2>&1; grep -E "passed|failed|error" $O/r05l_pytest.log | tail -3; grep -B5 -A30 "^E " $O/r05l_pytest.log | head -60
MLF_AB_ROUNDS=3 timeout 600 python scripts/sweep_ab.py 40 \
  filter_first_range_pct=30,filter_second_range_pct=0 \
  filter_first_range_pct=20,filter_second_range_pct=50 \
  filter_first_range_pct=15,filter_second_range_pct=40 \
  filter_first_range_pct=20,filter_second_range_pct=45 \
  filter_first_range_pct=15,filter_second_range_pct=50 \
  filter_first_range_pct=25,filter_second_range_pct=55 \
  filter_first_range_pct=20,filter_second_range_pct=60 \
  filter_first_range_pct=12,filter_second_range_pct=35 \
  2>/dev/null > $O/r05l_three_range_ab.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r05l_three_range_ab.jsonl'):
    d=json.loads(l); print(d['setting'], d['ms_per_step'], d['filter_launch_ms'], round(sum(d['filter_launch_ms']),4), d['mask_equals_exact'], d['stats'].get('second_range_groups'), d['stats'].get('third_range_groups'), d['stats'].get('uncertain_queries'))
PY
