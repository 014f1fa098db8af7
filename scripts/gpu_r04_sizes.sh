#!/bin/bash
# round 4: size check, end-to-end runs and the other configurations on the final tree
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== size check (P = 8e6, N = 2e5)"; timeout 900 python scripts/big_batch_check.py > $O/big_batch.json 2> $O/big_batch.err; tail -6 $O/big_batch.json | cut -c1-300; tail -2 $O/big_batch.err | cut -c1-200
echo "== end-to-end runs"; timeout 600 python scripts/e2e_run.py > $O/e2e_run.log 2>&1; tail -3 $O/e2e_run.log | cut -c1-400
echo "== config bench"; timeout 900 python scripts/config_bench.py > $O/config_bench.json 2> $O/config_bench.err; tail -2 $O/config_bench.err | cut -c1-200; python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/config_bench.json'))
    print(str(d)[:1500])
except Exception as e:
    print("config_bench:", e)
PY
