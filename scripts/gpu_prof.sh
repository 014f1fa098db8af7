#!/bin/bash
# rocprofv3 kernel stats of the headline bench (no CPU leg)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu > $O/prof_stats.log 2>&1
head -16 $O/prof_stats/bench_kernel_stats.csv | cut -c1-150
find $O -size +8M -delete
