#!/bin/bash
# round 5: k_prep_sweep with its waves in two shifts (fused_first_range = 2) against all eight in step (1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
S="fused_first_range=1 fused_first_range=2 fused_first_range=0"
: > $O/r05q_stagger_ab.jsonl
run() { echo "{\"case\": \"$1\"}" >> $O/r05q_stagger_ab.jsonl; env $2 MLF_AB_ROUNDS=3 timeout 300 python scripts/sweep_ab.py 40 $S 2>/dev/null >> $O/r05q_stagger_ab.jsonl; }
run "C5 P=1000000" "MLF_AB_P=1000000"
run "C5 P=262144" "MLF_AB_P=262144"
run "C5 P=4000000" "MLF_AB_P=4000000"
run "N=2000 d=20" "MLF_AB_N=2000 MLF_AB_D=20"
python - <<'PY'
import json
for l in open('gpurun_out/r05q_stagger_ab.jsonl'):
    d=json.loads(l)
    if 'case' in d: print('==', d['case']); continue
    print(d['setting'], d['ms_per_step'], d['filter_launch_ms'], d['mask_equals_exact'], d['stats'].get('range_cuts'))
PY
