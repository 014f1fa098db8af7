#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== tests"
timeout 900 python -m pytest tests/test_popstepsampler.py tests/test_philox.py tests/test_harness.py -m gpu -x -q > $O/r06g_tests.log 2>&1; tail -15 $O/r06g_tests.log | cut -c1-300
echo "== e2e"
timeout 600 python scripts/e2e_run.py nsteps10=40 > $O/r06g_e2e.log 2>&1; cut -c1-330 $O/r06g_e2e.log | tail -3
cp $O/e2e_run.json $O/r06g_e2e_run.json 2>/dev/null
echo "== refill"
timeout 300 python scripts/refill_profile.py 10 > $O/r06g_refill.json 2> $O/r06g_refill.err; cat $O/r06g_refill.json; tail -2 $O/r06g_refill.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06g_refill_stats -o st -- python $R/scripts/refill_profile.py 10 > $O/r06g_refill_stats.log 2>&1
head -10 $(find $O/r06g_refill_stats -name "*kernel_stats.csv" | head -1) | cut -c1-150
cp $(find $O/r06g_refill_stats -name "*kernel_stats.csv" | head -1) $O/r06g_refill_kernel_stats.csv
echo "== e2e under rocprofv3 (kernel statistics)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06g_e2e_stats -o st -- python $R/scripts/e2e_run.py nsteps10=40 > $O/r06g_e2e_stats.log 2>&1
head -14 $(find $O/r06g_e2e_stats -name "*kernel_stats.csv" | head -1) | cut -c1-170
cp $(find $O/r06g_e2e_stats -name "*kernel_stats.csv" | head -1) $O/r06g_e2e_kernel_stats.csv
find $O -name "*.csv" -size +4M -delete
