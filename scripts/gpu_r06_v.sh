#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 python $R/scripts/walk_large_profile.py 100000 25 2>/dev/null | cut -c1-300
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06v_walk -o st -- python $R/scripts/walk_large_profile.py 100000 25 > $O/r06v_walk.log 2>&1
head -12 $(find $O/r06v_walk -name "*kernel_stats.csv" | head -1) | cut -c1-150
find $O -name "*.csv" -size +4M -delete
cd $R
timeout 200 python scripts/walk_large_profile.py 10000 60 2>/dev/null | cut -c1-200
echo "== parity"; timeout 600 python -m pytest tests -m gpu -x -q -k "popstep or stepfunc or walk or harness" 2>&1 | tail -2
