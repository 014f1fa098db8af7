#!/bin/bash
# round 4, GPU call A: the sweep source / ceiling probe, the parity suite on the tree as it stands, the headline bench,
# and one PMC pass over the stage pipeline that separates "parked" (s_waitcnt) from "issue stall" time.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== probe"; timeout 300 scripts/probes/bin/sweep_src_probe > $O/r04_sweep_src_probe.json 2>&1; cat $O/r04_sweep_src_probe.json
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed\|Error" $O/pytest_gpu.log | tail -5
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json; tail -3 $O/bench.err
cd /tmp && export TMPDIR=/tmp
echo "== pmc wait split"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $O/r04a_pmc_wait -o st -- python $R/scripts/stage_profile.py 3 > $O/r04a_pmc_wait.log 2>&1
python $R/scripts/pmc_table.py $(find $O/r04a_pmc_wait -name "*counter_collection.csv" | head -1) > $O/r04a_pmc_wait.txt 2>&1
cat $O/r04a_pmc_wait.txt
find $O/r04a_pmc_wait -size +4M -delete
