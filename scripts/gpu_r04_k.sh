#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== filter / size / bounds tests"; timeout 900 python -m pytest tests/test_gpu_filter.py tests/test_config_sizes.py tests/test_regions.py tests/test_small_path.py tests/test_prep4_bounds.py -m gpu -x -q > $O/pytest_k.log 2>&1; grep -n "passed\|failed\|Error" $O/pytest_k.log | tail -5; tail -30 $O/pytest_k.log | grep -v "^$" | cut -c1-200 | head -30
echo "== A/B"
MLF_AB_ROUNDS=3 timeout 600 python scripts/sweep_ab.py 30 prep_direct=0 prep_direct=1 > $O/r04k3_ab.jsonl 2> $O/r04k3_ab.err
python - <<'PY'
import json
for l in open('gpurun_out/r04k3_ab.jsonl'):
    d=json.loads(l); print(d["setting"], d["ms_per_step"], d["filter_launch_ms"], d["mask_equals_exact"])
PY
tail -3 $O/r04k3_ab.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04k_stats -o st -- python $R/scripts/stage_profile.py 20 prep_direct=1 > $O/r04k_stats.log 2>&1
head -7 $(find $O/r04k_stats -name "*kernel_stats.csv" | head -1) | cut -c1-160
find $O/r04k_stats -size +4M -delete
