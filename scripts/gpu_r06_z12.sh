#!/bin/bash
# PMC pass over the refill kernels (f2): what bounds k_generate_ellipsoid / k_rows_affine
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $O/r06z12_pmc -o st -- python $R/scripts/refill_profile.py 3 --all > $O/r06z12_pmc.log 2>&1
python $R/scripts/pmc_table.py $(find $O/r06z12_pmc -name "*counter_collection.csv" | head -1) > $O/r06z12_pmc.txt 2>&1
grep -i "generate\|rows_affine\|scatter\|loglike\|Kernel\|kernel" $O/r06z12_pmc.txt | head -20 | cut -c1-260
for C in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/r06z12_pmc_$C -o st -- python $R/scripts/refill_profile.py 3 --all > $O/r06z12_pmc_$C.log 2>&1
python $R/scripts/pmc_table.py $(find $O/r06z12_pmc_$C -name "*counter_collection.csv" | head -1) > $O/r06z12_pmc_$C.txt 2>&1
grep -i "generate\|rows_affine" $O/r06z12_pmc_$C.txt | head -6 | cut -c1-200
done
find $O -name "*.csv" -size +4M -delete
