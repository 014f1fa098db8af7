#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -x -q > $O/r05f_pytest.log 2>&1; grep -E "passed|failed|error" $O/r05f_pytest.log | tail -3; grep -B5 -A30 "^E " $O/r05f_pytest.log | head -60
echo "== headline (k_prep4 reordered)"; timeout 600 python bench.py --headline-only --no-cpu --steps 20 --warmup 5 > $O/r05f_bench_headline.json 2>$O/r05f_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05f_bench_headline.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernel_ms'].items() if k in ('prep','scan','tail')}, d['roofline']['ms_per_launch_by_phase'], d['roofline']['frac'], d['roofline_prep']['achieved'])
PY
echo "== kernel trace of the d = 100 configuration"
export TMPDIR=/tmp
rm -rf /tmp/prof_d100; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_d100 -o d100 -- python $R/scripts/config_bench.py C5-at-d100 > /dev/null 2>&1)
f=$(find /tmp/prof_d100 -name "*kernel_stats.csv" | head -1); echo "$f"; cp "$f" $O/r05_d100_kernel_stats.csv; head -14 "$f" | cut -c1-220
