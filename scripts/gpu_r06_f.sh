#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== walker tests"
timeout 900 python -m pytest tests/test_popstepsampler.py tests/test_harness.py tests/test_config_sizes.py -m gpu -x -q > $O/r06f_tests.log 2>&1; tail -15 $O/r06f_tests.log | cut -c1-300
echo "== e2e"
timeout 600 python scripts/e2e_run.py nsteps10=40,80 > $O/r06f_e2e.log 2>&1; cut -c1-330 $O/r06f_e2e.log | tail -4
cp $O/e2e_run.json $O/r06f_e2e_run.json 2>/dev/null
echo "== fused A/B (one process)"
timeout 600 python scripts/fused_ab.py 200 loop:8:0 dma:8:1 dma_wave_slots:8:3 > $O/r06f_fused_ab.jsonl 2> $O/r06f_fused_ab.err; cut -c1-400 $O/r06f_fused_ab.jsonl; tail -3 $O/r06f_fused_ab.err
echo "== mid-size (k_prep4 tables by DMA)"
timeout 300 python scripts/midsize_profile.py > $O/r06f_midsize.json 2> $O/r06f_midsize.err; cat $O/r06f_midsize.json | cut -c1-160
echo "== refill"
timeout 300 python scripts/refill_profile.py 10 > $O/r06f_refill.json 2> $O/r06f_refill.err; cat $O/r06f_refill.json; tail -2 $O/r06f_refill.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06f_refill_stats -o st -- python $R/scripts/refill_profile.py 10 > $O/r06f_refill_stats.log 2>&1
head -16 $(find $O/r06f_refill_stats -name "*kernel_stats.csv" | head -1) | cut -c1-150
cp $(find $O/r06f_refill_stats -name "*kernel_stats.csv" | head -1) $O/r06f_refill_kernel_stats.csv
find $O -name "*.csv" -size +4M -delete
