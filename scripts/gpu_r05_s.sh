#!/bin/bash
# round 5: tile split of the single sweep (batches below the phased path): floor(slots / waves) against the cost rule
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
: > $O/r05s_tile_split_ab.jsonl
for p in 98304 140000 150000 163840 180000 196608 210000; do
  for rule in 0 1; do
    echo "{\"case\": \"C5 P=$p rule=$rule\"}" >> $O/r05s_tile_split_ab.jsonl
    MLF_SPLIT_RULE=$rule MLF_AB_P=$p MLF_AB_ROUNDS=2 timeout 300 python scripts/sweep_ab.py 60 filter=1 2>/dev/null >> $O/r05s_tile_split_ab.jsonl
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r05s_tile_split_ab.jsonl'):
    d=json.loads(l)
    if 'case' in d: print('==', d['case']); continue
    print('  ', d['ms_per_step'], d['filter_launch_ms'], d['mask_equals_exact'], d['stats'].get('segments'))
PY
