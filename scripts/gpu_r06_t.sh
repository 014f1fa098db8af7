#!/bin/bash
# experiment: k_sweep_min with THREE query groups per wave and tiles one ahead (156 registers: three waves per SIMD) against the
# production shape (four groups, tiles two ahead, 209 registers, two waves per SIMD); also 3 groups / two ahead, 4 groups / one ahead
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp ultranest_amd/libmlfriends_hip.so /tmp/new.so
cp scripts/probes/bin/libmlfriends_qw.so ultranest_amd/libmlfriends_hip.so
for Q in 0 3 32 41 0 3; do
echo "== MLF_SWEEP_QW=$Q"; MLF_SWEEP_QW=$Q timeout 200 python scripts/fused_ab.py 150 w4:4:1 2>/dev/null | head -2 | tail -1 | cut -c1-180
done
cp /tmp/new.so ultranest_amd/libmlfriends_hip.so
