#!/bin/bash
# round 5: k_prep_sweep with its waves walking the sets (one workgroup per CU: fused_first_range = 1) against one set per wave
# (fused_first_range = 2, the same kernel on the old grid); parity tests first
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_filter.py tests/test_config_sizes.py -m gpu -x -q > $O/r05z2_pytest.log 2>&1; grep -E "passed|failed|error" $O/r05z2_pytest.log | tail -3; grep -B5 -A30 "^E " $O/r05z2_pytest.log | head -60
MLF_AB_ROUNDS=3 timeout 300 python scripts/sweep_ab.py 40 fused_first_range=1 fused_first_range=2 2>/dev/null > $O/r05z2_walk_ab.jsonl
MLF_AB_P=524288 MLF_AB_ROUNDS=2 timeout 300 python scripts/sweep_ab.py 40 fused_first_range=1 fused_first_range=2 2>/dev/null >> $O/r05z2_walk_ab.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r05z2_walk_ab.jsonl'):
    d=json.loads(l); print(d['setting'], d['ms_per_step'], d['filter_launch_ms'], d['mask_equals_exact'])
PY
