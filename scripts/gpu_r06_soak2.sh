#!/bin/bash
# long soak on the final tree: 24 seed offsets of the random-shape tests, 8 of the full-size cases (incl. the same-form test)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
SOAK_A="801 802 803 804 805 806 807 808 809 810 811 812 813 814 815 816 817 818 819 820 821 822 823 824" SOAK_B="81 82 83 84 85 86 87 88" bash scripts/gpu_soak.sh 2>&1 | sort | uniq -c
cp gpurun_out/soak.log gpurun_out/r06_soak_long.log
