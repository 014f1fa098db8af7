#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== fused A/B, set E"
timeout 300 python scripts/fused_ab.py 200 w8:8:1 w4:4:1 > $O/r06j_fused_ab_E.jsonl 2> $O/r06j_fused_ab.err; cut -c1-230 $O/r06j_fused_ab_E.jsonl; tail -2 $O/r06j_fused_ab.err
echo "== fused A/B, set U"
MLF_AB_SET=U timeout 300 python scripts/fused_ab.py 200 w8:8:1 w4:4:1 > $O/r06j_fused_ab_U.jsonl 2> $O/r06j_fused_ab.err; cut -c1-230 $O/r06j_fused_ab_U.jsonl; tail -2 $O/r06j_fused_ab.err
