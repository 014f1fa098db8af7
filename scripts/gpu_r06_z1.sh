#!/bin/bash
# f2: fused ellipsoid generator, coalesced compaction, refill without the dense compaction: parity tests, refill profile, kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== tests"; timeout 900 python -m pytest tests/test_philox.py tests/test_harness.py -m gpu -x -q > $O/r06z1_tests.log 2>&1; tail -3 $O/r06z1_tests.log; grep -B5 -A25 "^E " $O/r06z1_tests.log | head -60
echo "== refill"; timeout 300 python scripts/refill_profile.py 10 > $O/r06z1_refill.json 2> $O/r06z1_refill.err; cat $O/r06z1_refill.json; tail -2 $O/r06z1_refill.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06z1_stats -o st -- python $R/scripts/refill_profile.py 10 > $O/r06z1_stats.log 2>&1
head -22 $(find $O/r06z1_stats -name "*kernel_stats.csv" | head -1) | cut -c1-150
cp $(find $O/r06z1_stats -name "*kernel_stats.csv" | head -1) $O/r06z1_refill_kernel_stats.csv
cd $R
echo "== sample bench"; timeout 300 python scripts/sample_bench.py > $O/r06z1_sample_bench.json 2> $O/r06z1_sample_bench.err; tail -c 700 $O/r06z1_sample_bench.json
