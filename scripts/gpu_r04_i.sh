#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== mid-size (one launch)"; timeout 300 python scripts/midsize_profile.py > $O/midsize.json 2> $O/midsize.err; cat $O/midsize.json | cut -c1-300; tail -3 $O/midsize.err
echo "== mid-size (three launches)"; timeout 300 python scripts/midsize_profile.py mid_max_queries=0 > $O/midsize3.json 2> $O/midsize3.err; cat $O/midsize3.json | cut -c1-200
