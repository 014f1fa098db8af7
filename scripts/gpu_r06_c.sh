#!/bin/bash
# round 6, call c: k_prep_sweep after the table-load change + finer stamps; bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== fused A/B"
timeout 300 python scripts/fused_ab.py 300 > $O/r06c_fused_ab.jsonl 2> $O/r06c_fused_ab.err; cut -c1-900 $O/r06c_fused_ab.jsonl; tail -3 $O/r06c_fused_ab.err
echo "== filter tests"
timeout 900 python -m pytest tests/test_config_sizes.py tests/test_gpu_filter.py -m gpu -x -q 2>&1 | tail -3
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06c_bench.json 2> $O/r06c_bench.err; tail -3 $O/r06c_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06c_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('rebuild_ms'), d.get('rebuild_device_resident_ms'), d['strong_scaling']['ms_per_step'])
r=d['roofline']; print({k:r.get(k) for k in ('kernel','bound','achieved','frac','traffic','traffic_status','ms_per_launch')})
print([(e['kernel'], round(e['ms'],4), round(e['mfma_frac_of_2500'],3), round(e['hbm_frac_of_8000'],3)) for e in r['launches']])
print(r['step']); print(r['fp64_valu_roofline_of_survey_8d'] and r['fp64_valu_roofline_of_survey_8d']['ratio_to_that_peak'])
print(d['kernel_ms']['wall_ms_per_step_of_the_launch_event_pass'], d['cpu_baseline']['value'], d['cpu_baseline']['gpu_mask_equals_cpu_mask_on_sample'])
PY
