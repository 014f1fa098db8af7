#!/bin/bash
# rounds of the population slice sampler: ring walker and followers in one launch (hand-shake) -- parity, one-call profile, C3 end to end
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== parity"; timeout 600 python -m pytest tests -m gpu -x -q -k "popstep or stepfunc or walk or harness" 2>&1 | tail -3
echo "== one call"; timeout 200 python scripts/walk_rounds_profile.py 1500 2>/dev/null | tee $O/r06_walk_rounds.json
echo "== walk bench"; timeout 300 python scripts/walk_bench.py > $O/r06_walk_bench.json 2> $O/r06_walk_bench.err; tail -c 700 $O/r06_walk_bench.json
echo "== end to end"; timeout 300 python scripts/e2e_run.py nsteps10=40,80 > $O/r06_e2e_run.log 2>&1; tail -2 $O/r06_e2e_run.log | cut -c1-330; cp $O/e2e_run.json $O/r06_e2e_run.json
