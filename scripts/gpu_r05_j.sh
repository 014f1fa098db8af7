#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== durations"; timeout 900 python -m pytest tests/test_config_sizes.py -m gpu -q -k "optional_routings" --durations=12 2>&1 | tail -22 | cut -c1-200
echo "== d 65..128 and wide tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "wide or prep64" --durations=5 2>&1 | tail -25 | cut -c1-220
echo "== d = 100"; timeout 600 python scripts/config_bench.py C5-at-d100 published-shape 2>/dev/null | python -c "
import json,sys
c=json.load(sys.stdin)
for k,v in c.items():
    if isinstance(v,dict) and 'E' in v: print(k, v['per_proposal_stage'][:40], 'E %.3g U %.3g F %.3g N %.3g'%tuple(v[s]['proposals_per_s'] for s in 'EUFN'), v.get('rebuild_ms'))
"
