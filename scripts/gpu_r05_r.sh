#!/bin/bash
# round 5: where the phased (min-only) path should start: batches of 65536 ... 262144 through the single sweep (default
# below 3e7 proposals x tiles) against the two-range path, with the per-proposal stage inside and in front of the first launch
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
S="filter_phase_min_queries=32768,fused_first_range=1 filter_phase_min_queries=32767,fused_first_range=1 filter_phase_min_queries=32767,fused_first_range=0"
: > $O/r05r_phase_threshold_ab.jsonl
run() { echo "{\"case\": \"$1\"}" >> $O/r05r_phase_threshold_ab.jsonl; env $2 MLF_AB_ROUNDS=3 timeout 300 python scripts/sweep_ab.py 60 $S 2>/dev/null >> $O/r05r_phase_threshold_ab.jsonl; }
for p in 65536 98304 131072 163840 196608 229376 262144; do run "C5 P=$p" "MLF_AB_P=$p"; done
python - <<'PY'
import json
for l in open('gpurun_out/r05r_phase_threshold_ab.jsonl'):
    d=json.loads(l)
    if 'case' in d: print('==', d['case']); continue
    print(d['setting'], d['ms_per_step'], d['filter_launch_ms'], d['mask_equals_exact'])
PY
