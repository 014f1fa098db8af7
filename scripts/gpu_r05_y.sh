#!/bin/bash
# round 5: k_sweep_min walking a compacted set with a capped grid -- the multi-pass case (8e6 and 4e6 proposals, set N)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python scripts/big_batch_check.py > $O/big_batch.json 2> $O/big_batch.err; cat $O/big_batch.json | tr -d '\n' | cut -c1-600; echo
MLF_AB_P=4000000 MLF_AB_ROUNDS=2 timeout 300 python scripts/sweep_ab.py 30 filter=1 2>/dev/null | cut -c1-200
MLF_AB_R2_SCALE=0.2 MLF_AB_ROUNDS=2 timeout 300 python scripts/sweep_ab.py 30 filter=1 2>/dev/null | cut -c1-200
