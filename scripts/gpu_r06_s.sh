#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python scripts/sweep_stamps.py 2>/dev/null | tee $O/r06_sweep_min_stamps.jsonl
timeout 200 python scripts/fused_ab.py 150 w4:4:1 2>/dev/null | head -2 | tail -1 | cut -c1-180
