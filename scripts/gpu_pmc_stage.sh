#!/bin/bash
# PMC passes over the C5 membership pipeline (scripts/stage_profile.py): instruction mix / matrix pipe, then HBM traffic.
# usage: scripts/gpu_pmc_stage.sh <tag> [option=value ...]      results: gpurun_out/<tag>_pmc_*.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/${TAG}_pmc_sq -o st -- python $R/scripts/stage_profile.py 3 "$@" > $O/${TAG}_pmc_sq.log 2>&1
python $R/scripts/pmc_table.py $(find $O/${TAG}_pmc_sq -name "*counter_collection.csv" | head -1) > $O/${TAG}_pmc_sq.txt 2>&1
if [ "$PMC_TRAFFIC" = "1" ]; then
for C in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${TAG}_pmc_$C -o st -- python $R/scripts/stage_profile.py 3 "$@" > $O/${TAG}_pmc_$C.log 2>&1
python $R/scripts/pmc_table.py $(find $O/${TAG}_pmc_$C -name "*counter_collection.csv" | head -1) > $O/${TAG}_pmc_$C.txt 2>&1
done
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o st -- python $R/scripts/stage_profile.py 10 "$@" > $O/${TAG}_stats.log 2>&1
find $O/${TAG}_pmc_sq $O/${TAG}_stats -size +4M -delete
cat $O/${TAG}_pmc_sq.txt
head -12 $(find $O/${TAG}_stats -name "*kernel_stats.csv" | head -1)
