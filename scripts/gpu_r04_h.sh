#!/bin/bash
# round 4, GPU call H: the whole GPU suite, smoke, bench, reference grid, mid-size, PMC on the stage pipeline
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed\|Error" $O/pytest_gpu.log | tail -5
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('rebuild_ms'), d.get('rebuild_device_resident_ms'))
r=d['roofline']; print({k:r[k] for k in ('achieved','frac','ms_per_launch_by_phase','executed_mfma_per_launch_by_phase','achieved_by_phase','uncertain_queries','uncertain_pairs')})
print(d.get('kernel_ms'))
print(d['cpu_baseline'])
PY
tail -3 $O/bench.err
echo "== reference grid"; timeout 600 python scripts/reference_grid_bench.py > $O/reference_grid.json 2> $O/reference_grid.err; tail -5 $O/reference_grid.err | cut -c1-300
echo "== mid-size"; timeout 300 python scripts/midsize_profile.py > $O/midsize.json 2> $O/midsize.err; cat $O/midsize.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
echo "== pmc"
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/r04h_pmc_sq -o st -- python $R/scripts/stage_profile.py 3 > $O/r04h_pmc_sq.log 2>&1
python $R/scripts/pmc_table.py $(find $O/r04h_pmc_sq -name "*counter_collection.csv" | head -1) > $O/r04h_pmc_sq.txt 2>&1
grep -A9 "k_sweep_min\|k_uncertain" $O/r04h_pmc_sq.txt | head -60
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $O/r04h_pmc_wait -o st -- python $R/scripts/stage_profile.py 3 > $O/r04h_pmc_wait.log 2>&1
python $R/scripts/pmc_table.py $(find $O/r04h_pmc_wait -name "*counter_collection.csv" | head -1) > $O/r04h_pmc_wait.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/r04h_pmc_$C -o st -- python $R/scripts/stage_profile.py 3 > $O/r04h_pmc_$C.log 2>&1
python $R/scripts/pmc_table.py $(find $O/r04h_pmc_$C -name "*counter_collection.csv" | head -1) > $O/r04h_pmc_$C.txt 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04h_stats -o st -- python $R/scripts/stage_profile.py 20 > $O/r04h_stats.log 2>&1
head -8 $(find $O/r04h_stats -name "*kernel_stats.csv" | head -1) | cut -c1-160
find $O -name "*.csv" -size +4M -delete
