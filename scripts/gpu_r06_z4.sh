#!/bin/bash
# timeline of one steady-state rebuild: kernels, copies, gaps
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/r06z4_trace -o st -- python $R/scripts/rebuild_trace.py 6 > $O/r06z4_trace.log 2>&1
python $R/scripts/rebuild_timeline.py $O/r06z4_trace > $O/r06z4_timeline.txt 2>&1; cat $O/r06z4_timeline.txt | tail -80
cd $R; timeout 200 python scripts/rebuild_calls.py 2>/dev/null | head -12
