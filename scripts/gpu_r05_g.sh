#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace of the d = 100 configuration"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05_d100_stats -o st -- python $R/scripts/config_bench.py C5-at-d100 > $O/r05_d100.log 2>&1
f=$(find $O/r05_d100_stats -name "*kernel_stats.csv" | head -1); echo "$f"; cp "$f" $O/r05_d100_kernel_stats.csv; head -16 "$f" | cut -c1-200
cd $R
echo "== headline"; timeout 600 python bench.py --headline-only --no-cpu --steps 20 --warmup 5 > $O/r05g_bench_headline.json 2>$O/r05g_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05g_bench_headline.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernel_ms'].items() if k in ('prep','scan','tail')}, d['roofline']['ms_per_launch_by_phase'], d['roofline']['frac'], d['roofline_prep']['achieved'])
PY
find $O -name "*.csv" -size +4M -delete
