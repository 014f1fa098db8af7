#!/bin/bash
# round 5: mid-size batches on the final tree (the measuring script now discards its first pass)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python scripts/midsize_profile.py > $O/midsize.json 2> $O/midsize.err; cat $O/midsize.json | cut -c1-120
