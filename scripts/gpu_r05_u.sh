#!/bin/bash
# round 5: size check (8e6 proposals; 2e5 live points) and a longer soak on the final tree
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python scripts/big_batch_check.py > $O/big_batch.json 2> $O/big_batch.err; cat $O/big_batch.json | cut -c1-900; tail -2 $O/big_batch.err
SOAK_A="601 602 603 604 605 606 607 608 609 610 611 612 613 614 615 616" SOAK_B="61 62 63 64 65" bash scripts/gpu_soak.sh 2>&1 | tail -22
