#!/bin/bash
# cube test of device-side sampling inside the fused first launch; one count + scan per mask: parity, refill profile
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== tests"; timeout 900 python -m pytest tests/test_philox.py tests/test_harness.py tests/test_regions.py -m gpu -x -q > $O/r06z6_tests.log 2>&1; tail -3 $O/r06z6_tests.log; grep -B5 -A25 "^E " $O/r06z6_tests.log | head -60
echo "== refill"; timeout 300 python scripts/refill_profile.py 10 --all > $O/r06z6_refill.json 2> $O/r06z6_refill.err; cat $O/r06z6_refill.json; tail -2 $O/r06z6_refill.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06z6_stats -o st -- python $R/scripts/refill_profile.py 10 --all > $O/r06z6_stats.log 2>&1
cp $(find $O/r06z6_stats -name "*kernel_stats.csv" | head -1) $O/r06z6_refill_kernel_stats.csv
