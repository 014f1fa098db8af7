#!/bin/bash
# round 5: the whole GPU suite on the tree with three ranges / fused first launch / cost-rule tile split as defaults, then the
# share of the first of TWO ranges (batches of 240 000 ... 800 000 proposals) with the per-proposal stage inside it
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/r05t_pytest.log 2>&1; grep -E "passed|failed|error" $O/r05t_pytest.log | tail -3; grep -B5 -A30 "^E " $O/r05t_pytest.log | head -80; grep -A10 "slowest" $O/r05t_pytest.log | head -12
S=""
for a in 20 25 30 35 40; do S="$S filter_first_range_pct=$a"; done
: > $O/r05t_two_range_first_pct.jsonl
for p in 300000 524288 700000; do
  echo "{\"case\": \"C5 P=$p\"}" >> $O/r05t_two_range_first_pct.jsonl
  MLF_AB_P=$p MLF_AB_ROUNDS=2 timeout 300 python scripts/sweep_ab.py 40 $S 2>/dev/null >> $O/r05t_two_range_first_pct.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r05t_two_range_first_pct.jsonl'):
    d=json.loads(l)
    if 'case' in d: print('==', d['case']); continue
    print('  ', d['setting'], d['ms_per_step'], d['filter_launch_ms'], d['mask_equals_exact'], d['stats'].get('range_cuts'))
PY
