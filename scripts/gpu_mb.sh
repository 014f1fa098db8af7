#!/bin/bash
# microbench + kernel stats (quick look at per-kernel times of the inside pipeline)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_mb
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mb -o mb -- python $R/scripts/microbench.py quick > $O/microbench.log 2>&1
grep inside $O/microbench.log | cut -c1-330
head -9 $O/prof_mb/mb_kernel_stats.csv | cut -c1-120
find $O -size +8M -delete
