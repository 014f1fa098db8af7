#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python scripts/range_cut_sweep.py 150 2>/dev/null > $O/r06_range_cut_sweep.jsonl; tail -1 $O/r06_range_cut_sweep.jsonl; grep '"first": 30, "second": 50' $O/r06_range_cut_sweep.jsonl
