#!/bin/bash
# round 4, GPU call E: min-only sweep -- parity (whole GPU suite, default = new path; filter tests again with sweep_min=0), A/B, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== pytest -m gpu (sweep_min = 1, default)"; timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed\|Error" $O/pytest_gpu.log | tail -5
echo "== filter / size tests with sweep_min=0"
MLF_TEST_OPTIONS=sweep_min=0 timeout 900 python -m pytest tests/test_gpu_filter.py tests/test_config_sizes.py -m gpu -x -q > $O/pytest_min0.log 2>&1; grep -n "passed\|failed\|Error" $O/pytest_min0.log | tail -5
echo "== A/B"
MLF_AB_ROUNDS=3 timeout 600 python scripts/sweep_ab.py 30 sweep_min=0 sweep_min=1 > $O/r04e_ab.jsonl 2> $O/r04e_ab.err; cut -c1-420 $O/r04e_ab.jsonl; tail -3 $O/r04e_ab.err
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('rebuild_ms'), d.get('rebuild_device_resident_ms'))
r=d['roofline']; print({k:r[k] for k in ('achieved','frac','ms_per_launch_by_phase','executed_mfma_per_launch_by_phase','achieved_by_phase','uncertain_queries','uncertain_pairs')})
print(d.get('kernel_ms'))
PY
tail -3 $O/bench.err
