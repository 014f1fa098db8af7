#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== fused A/B (one process)"
timeout 600 python scripts/fused_ab.py 200 base:8:0 dma:8:1 stag64:8:$((2 + 64*256)) stag128:8:$((2 + 128*256)) stag192:8:$((2 + 192*256)) dma_stag128:8:$((3 + 128*256)) w4:4:0 w4dma:4:1 > $O/r06d_fused_ab.jsonl 2> $O/r06d_fused_ab.err; cut -c1-1000 $O/r06d_fused_ab.jsonl; tail -3 $O/r06d_fused_ab.err
