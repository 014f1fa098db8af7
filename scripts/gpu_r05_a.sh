#!/bin/bash
# round 5, call a: parity of the centre-first mask-mode operand + A/B of the first range's share
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== gpu tests"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/r05a_pytest.log
echo "== order / first range A/B"; timeout 600 python scripts/sweep_ab.py 20 filter_order=0,filter_first_range_pct=50 filter_order=1,filter_first_range_pct=50 \
  filter_order=1,filter_first_range_pct=40 filter_order=1,filter_first_range_pct=35 filter_order=1,filter_first_range_pct=30 filter_order=1,filter_first_range_pct=25 \
  filter_order=1,filter_first_range_pct=20 filter_order=0,filter_first_range_pct=40 > $O/r05a_order_ab.jsonl 2> $O/r05a_order_ab.err
cut -c1-330 $O/r05a_order_ab.jsonl; tail -3 $O/r05a_order_ab.err
