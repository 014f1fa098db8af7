#!/bin/bash
# k_boot_sym: parity (K4 tests), A/B against k_boot, kernel durations
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== K4 tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "bootstrap" > $O/pytest_n.log 2>&1; grep -n "passed\|failed\|Error" $O/pytest_n.log | tail -5; tail -30 $O/pytest_n.log | grep -v "^$" | cut -c1-200 | head -30
echo "== A/B"; timeout 300 python scripts/boot_ab.py > $O/r04_boot_ab.json 2> $O/r04_boot_ab.err; cat $O/r04_boot_ab.json; tail -3 $O/r04_boot_ab.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04n_stats -o st -- python $R/scripts/boot_ab.py > $O/r04n_stats.log 2>&1
head -8 $(find $O/r04n_stats -name "*kernel_stats.csv" | head -1) | cut -c1-160
find $O/r04n_stats -size +4M -delete
