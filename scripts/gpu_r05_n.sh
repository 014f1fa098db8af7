#!/bin/bash
# round 5: three ranges with the per-proposal stage inside the first -- grid of the first cut
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
S="filter_first_range_pct=30,filter_second_range_pct=0,fused_first_range=0 filter_first_range_pct=15,filter_second_range_pct=50,fused_first_range=0"
for a in 10 12 15 18 20 22; do for b in 50 55; do S="$S filter_first_range_pct=$a,filter_second_range_pct=$b,fused_first_range=1"; done; done
MLF_AB_ROUNDS=2 timeout 900 python scripts/sweep_ab.py 40 $S 2>/dev/null > $O/r05n_three_range_fused_grid.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r05n_three_range_fused_grid.jsonl'):
    d=json.loads(l); print(d['setting'].replace('filter_','').replace('_range_pct',''), d['ms_per_step'], d['filter_launch_ms'], round(sum(d['filter_launch_ms']),4), d['mask_equals_exact'])
PY
