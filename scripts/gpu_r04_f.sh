#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== filter / size tests"; timeout 900 python -m pytest tests/test_gpu_filter.py tests/test_config_sizes.py -m gpu -x -q > $O/pytest_f.log 2>&1; grep -n "passed\|failed\|Error" $O/pytest_f.log | tail -5
echo "== A/B"
MLF_AB_ROUNDS=2 timeout 600 python scripts/sweep_ab.py 30 sweep_min=0 sweep_min=1 > $O/r04f_ab.jsonl 2> $O/r04f_ab.err
python - <<'PY'
import json
for l in open('gpurun_out/r04f_ab.jsonl'):
    d=json.loads(l); print(d["setting"], d["ms_per_step"], d["filter_launch_ms"], d["mask_equals_exact"], d["stats"].get("uncertain_stage_cycles"), d["stats"].get("uncertain_queries"), d["stats"].get("uncertain_pairs"))
PY
tail -3 $O/r04f_ab.err
cd /tmp && export TMPDIR=/tmp
echo "== kernel stats"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04f_stats -o st -- python $R/scripts/stage_profile.py 20 > $O/r04f_stats.log 2>&1
head -7 $(find $O/r04f_stats -name "*kernel_stats.csv" | head -1) | cut -c1-200
find $O/r04f_stats -size +4M -delete
