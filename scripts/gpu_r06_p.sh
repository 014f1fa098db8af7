#!/bin/bash
# k_uncertain with sweeping and gathering waves (fused_variant bit 2) against the one-chain form, inside ONE process; stage words
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python scripts/fused_ab.py 200 chain:4:1 split:4:5 2>$O/r06_p.err | tee $O/r06_uncertain_split_ab.jsonl | cut -c1-200
echo "== words, chain"; MLF_VARIANT=1 timeout 100 python scripts/uncertain_probe.py 2>/dev/null | tail -2
echo "== words, split"; MLF_VARIANT=5 timeout 100 python scripts/uncertain_probe.py 2>/dev/null | tail -2
