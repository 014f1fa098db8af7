#!/bin/bash
# the range cuts once more, with the cheaper first launch (same quadratic form): first share 25 / 30 / 35 % of the second cut 45 / 50 / 55 %
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
MLF_AB_ROUNDS=2 timeout 600 python scripts/sweep_ab.py 100 \
  filter_first_range_pct=30,filter_second_range_pct=50 filter_first_range_pct=25,filter_second_range_pct=50 filter_first_range_pct=35,filter_second_range_pct=50 \
  filter_first_range_pct=30,filter_second_range_pct=45 filter_first_range_pct=30,filter_second_range_pct=55 filter_first_range_pct=35,filter_second_range_pct=55 \
  filter_first_range_pct=25,filter_second_range_pct=45 filter_first_range_pct=40,filter_second_range_pct=50 filter_first_range_pct=30,filter_second_range_pct=60 \
  > $O/r06z11_cuts.jsonl 2> $O/r06z11.err
python - <<'PY'
import json
for l in open('gpurun_out/r06z11_cuts.jsonl'):
    try: d=json.loads(l)
    except Exception: continue
    print({k:d[k] for k in d if k in ('setting','ms_per_step','launch_ms','mask_equals_exact','masks_equal_exact')})
PY
tail -2 $O/r06z11.err
