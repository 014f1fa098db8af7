#!/bin/bash
# HBM traffic of the membership pipeline with the per-proposal stage inside the first sweep launch (fused_first_range=1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/r04f_pmc_$C -o st -- python $R/scripts/stage_profile.py 3 fused_first_range=1 > $O/r04f_pmc_$C.log 2>&1
python $R/scripts/pmc_table.py $(find $O/r04f_pmc_$C -name "*counter_collection.csv" | head -1) > $O/r04f_pmc_$C.txt 2>&1
done
grep -A2 "k_prep_sweep\|k_sweep_min\|k_uncertain\|k_scan" $O/r04f_pmc_FETCH_SIZE.txt | head -30
grep -A2 "k_prep_sweep\|k_sweep_min\|k_uncertain\|k_scan" $O/r04f_pmc_WRITE_SIZE.txt | head -30
find $O -name "*.csv" -size +4M -delete
