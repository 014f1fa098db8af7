#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== filter / size / rebuild tests"; timeout 900 python -m pytest tests/test_gpu_filter.py tests/test_config_sizes.py tests/test_device_rebuild.py tests/test_gpu_kernels.py tests/test_distributed.py -m gpu -x -q > $O/pytest_g.log 2>&1; grep -n "passed\|failed\|Error" $O/pytest_g.log | tail -5; tail -30 $O/pytest_g.log | grep -v "^$" | head -40
echo "== A/B"
MLF_AB_ROUNDS=2 timeout 600 python scripts/sweep_ab.py 30 sweep_min=0 sweep_min=1 > $O/r04f_ab.jsonl 2> $O/r04f_ab.err
python - <<'PY'
import json
for l in open('gpurun_out/r04f_ab.jsonl'):
    d=json.loads(l); print(d["setting"], d["ms_per_step"], d["filter_launch_ms"], d["mask_equals_exact"], d["stats"].get("uncertain_stage_cycles"), d["stats"].get("uncertain_queries"), d["stats"].get("uncertain_pairs"))
PY
tail -3 $O/r04f_ab.err
echo "== rebuild modes"; timeout 300 python scripts/rebuild_modes.py > $O/rebuild_modes.json 2> $O/rebuild_modes.err; tail -c 1500 $O/rebuild_modes.json; tail -5 $O/rebuild_modes.err
