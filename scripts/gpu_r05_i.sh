#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -x -q > $O/r05i_pytest.log 2>&1; grep -E "passed|failed|error" $O/r05i_pytest.log | tail -3; grep -B5 -A30 "^E " $O/r05i_pytest.log | head -60
for i in 1 2; do
echo "== headline"; timeout 600 python bench.py --headline-only --no-cpu --steps 20 --warmup 5 > $O/r05i_bench_headline.json 2>$O/r05i_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05i_bench_headline.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernel_ms'].items() if k in ('prep','scan','tail')}, d['roofline']['ms_per_launch_by_phase'], d['roofline']['frac'], d['roofline_prep']['achieved'])
PY
done
echo "== d = 100"; timeout 600 python scripts/config_bench.py C5-at-d100 published-shape 2>/dev/null | python -c "
import json,sys
c=json.load(sys.stdin)
for k,v in c.items():
    if isinstance(v,dict) and 'E' in v: print(k, 'E %.3g U %.3g F %.3g N %.3g'%tuple(v[s]['proposals_per_s'] for s in 'EUFN'))
"
