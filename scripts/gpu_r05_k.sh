#!/bin/bash
# round 5: the GPU suite and the smoke test on the committed tree (last check)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r05k_pytest.log 2>&1; grep -E "passed|failed|error" $O/r05k_pytest.log | tail -3; grep -B5 -A30 "^E " $O/r05k_pytest.log | head -60
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
