#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== walk rounds profile"
timeout 200 python scripts/walk_rounds_profile.py 1500 > $O/r06h_walk_rounds.json 2> $O/r06h_walk_rounds.err; cat $O/r06h_walk_rounds.json; tail -2 $O/r06h_walk_rounds.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06h_walk_stats -o st -- python $R/scripts/walk_rounds_profile.py 600 > $O/r06h_walk_stats.log 2>&1
head -12 $(find $O/r06h_walk_stats -name "*kernel_stats.csv" | head -1) | cut -c1-200
cp $(find $O/r06h_walk_stats -name "*kernel_stats.csv" | head -1) $O/r06h_walk_kernel_stats.csv
find $O -name "*.csv" -size +4M -delete
