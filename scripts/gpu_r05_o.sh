#!/bin/bash
# round 5: two ranges against three (with and without the per-proposal stage inside the first) on other batch sizes, on a
# set with almost no neighbour within reach, and on another region shape
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
S="filter_first_range_pct=30,filter_second_range_pct=0,fused_first_range=0 filter_first_range_pct=15,filter_second_range_pct=50,fused_first_range=0 filter_first_range_pct=15,filter_second_range_pct=50,fused_first_range=1 filter_first_range_pct=30,filter_second_range_pct=0,fused_first_range=1"
: > $O/r05o_range_ab.jsonl
run() { echo "{\"case\": \"$1\"}" >> $O/r05o_range_ab.jsonl; env $2 MLF_AB_ROUNDS=2 timeout 300 python scripts/sweep_ab.py 30 $S 2>/dev/null >> $O/r05o_range_ab.jsonl; }
run "C5 P=262144" "MLF_AB_P=262144"
run "C5 P=524288" "MLF_AB_P=524288"
run "C5 P=4000000" "MLF_AB_P=4000000"
run "C5 set N" "MLF_AB_R2_SCALE=0.2"
run "C5 radius x 2" "MLF_AB_R2_SCALE=2"
run "N=2000 d=20" "MLF_AB_N=2000 MLF_AB_D=20"
run "N=10000 d=30" "MLF_AB_N=10000 MLF_AB_D=30"
run "N=1000 d=10" "MLF_AB_N=1000 MLF_AB_D=10"
python - <<'PY'
import json
for l in open('gpurun_out/r05o_range_ab.jsonl'):
    d=json.loads(l)
    if 'case' in d: print('==', d['case']); continue
    print(d['setting'].replace('filter_','').replace('_range_pct',''), d['ms_per_step'], d['filter_launch_ms'], d['mask_equals_exact'], d['accept'], d['stats'].get('second_range_groups'), d['stats'].get('third_range_groups'))
PY
