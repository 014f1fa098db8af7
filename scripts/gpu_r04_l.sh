#!/bin/bash
# k_uncertain with two of its eight waves fetching the set's rows next to the sweep: parity subset, headline, stage stamps
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== filter / size / region tests"; timeout 900 python -m pytest tests/test_gpu_filter.py tests/test_config_sizes.py tests/test_regions.py tests/test_small_path.py -m gpu -x -q > $O/pytest_l.log 2>&1; grep -n "passed\|failed\|Error" $O/pytest_l.log | tail -5; tail -30 $O/pytest_l.log | grep -v "^$" | cut -c1-200 | head -30
echo "== headline"
MLF_AB_ROUNDS=3 timeout 600 python scripts/sweep_ab.py 30 sweep_min=1 > $O/r04l_ab.jsonl 2> $O/r04l_ab.err
python - <<'PY'
import json
for l in open('gpurun_out/r04l_ab.jsonl'):
    d=json.loads(l); print(d["setting"], d["ms_per_step"], d["filter_launch_ms"], d["mask_equals_exact"])
PY
tail -3 $O/r04l_ab.err
echo "== stage stamps"; timeout 300 python scripts/midsize_profile.py time_filter_launches=1 131072 > $O/r04l_stamps.json 2> $O/r04l_stamps.err; cut -c1-600 $O/r04l_stamps.json; tail -3 $O/r04l_stamps.err
