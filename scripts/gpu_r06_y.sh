#!/bin/bash
# k_prep_sweep<.., SQ>: the ellipsoid form read off the whitening chain ("fused_variant" bit 1).  Parity first (the new full-size
# test, the filter / size tests), then variant 3 against variant 1 in ONE process (scripts/fused_ab.py), then the bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== same-form tests"; timeout 600 python -m pytest tests/test_config_sizes.py -m gpu -x -q -k "same_quadratic or full_size" > $O/r06y_tests.log 2>&1; tail -3 $O/r06y_tests.log; grep -B5 -A25 "^E " $O/r06y_tests.log | head -60
echo "== A/B"; timeout 400 python scripts/fused_ab.py 200 base:4:1 sq:4:3 > $O/r06y_fused_ab.jsonl 2> $O/r06y_fused_ab.err; cut -c1-200 $O/r06y_fused_ab.jsonl; tail -2 $O/r06y_fused_ab.err
echo "== filter tests"; timeout 600 python -m pytest tests/test_gpu_filter.py tests/test_regions.py tests/test_abi.py -m gpu -x -q > $O/r06y_tests2.log 2>&1; tail -3 $O/r06y_tests2.log; grep -B5 -A25 "^E " $O/r06y_tests2.log | head -40
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu > $O/r06y_bench.json 2> $O/r06y_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06y_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('rebuild_ms'))
r=d['roofline']; print([(e['kernel'], round(e['ms'],4)) for e in r['launches']])
PY
tail -3 $O/r06y_bench.err
