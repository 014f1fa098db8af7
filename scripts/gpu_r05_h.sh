#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -x -q > $O/r05h_pytest.log 2>&1; grep -E "passed|failed|error" $O/r05h_pytest.log | tail -3; grep -B5 -A30 "^E " $O/r05h_pytest.log | head -60
echo "== region set"; timeout 120 python scripts/region_set_breakdown.py 2>/dev/null | tail -6 | cut -c1-200
echo "== rebuild"; timeout 300 python scripts/rebuild_modes.py 2> /dev/null | head -4 | tr -d '\n' | cut -c1-200; echo
echo "== small batches"; timeout 300 python scripts/small_batch_latency.py 2>/dev/null | tail -4 | cut -c1-300
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05h_set_stats -o st -- python $R/scripts/region_set_breakdown.py > /dev/null 2>&1
grep -i "k_ref\|quant_refs\|k_whiten_rows\|build_layouts\|scatter_copy" $(find $O/r05h_set_stats -name "*kernel_stats.csv" | head -1) | cut -c1-140
find $O -name "*.csv" -size +4M -delete
