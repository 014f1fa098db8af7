#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== fused A/B (one process)"
timeout 300 python scripts/fused_ab.py 200 loop:8:0 dma:8:1 dma_il:8:5 w4:4:1 > $O/r06i_fused_ab.jsonl 2> $O/r06i_fused_ab.err; cut -c1-500 $O/r06i_fused_ab.jsonl; tail -3 $O/r06i_fused_ab.err
echo "== parity with the interleaved chains as the process default"
MLF_TEST_OPTIONS="fused_variant=5" timeout 600 python -m pytest tests/test_config_sizes.py tests/test_gpu_filter.py tests/test_prep4_bounds.py -m gpu -x -q 2>&1 | tail -3
