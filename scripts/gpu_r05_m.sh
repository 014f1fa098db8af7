#!/bin/bash
# round 5: min-only sweep in three ranges -- grid of the two cuts, and the per-proposal stage inside the first range
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
S="filter_first_range_pct=30,filter_second_range_pct=0,fused_first_range=0"
for a in 10 12 15 18; do for b in 40 45 50 55 60; do S="$S filter_first_range_pct=$a,filter_second_range_pct=$b,fused_first_range=0"; done; done
S="$S filter_first_range_pct=15,filter_second_range_pct=50,fused_first_range=1 filter_first_range_pct=10,filter_second_range_pct=45,fused_first_range=1 filter_first_range_pct=30,filter_second_range_pct=0,fused_first_range=1"
MLF_AB_ROUNDS=2 timeout 900 python scripts/sweep_ab.py 40 $S 2>/dev/null > $O/r05m_three_range_grid.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r05m_three_range_grid.jsonl'):
    d=json.loads(l); print(d['setting'].replace('filter_','').replace('_range_pct',''), d['ms_per_step'], d['filter_launch_ms'], round(sum(d['filter_launch_ms']),4), d['mask_equals_exact'])
PY
