#!/bin/bash
# SQ wait-state counters for the headline kernels (microbench quick run)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_wait
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_wait -o mb -- python $R/scripts/microbench.py quick > $O/pmc_wait.log 2>&1
tail -3 $O/pmc_wait.log | cut -c1-200
python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$O/pmc_wait/mb_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n=r["Kernel_Name"]
    if ("k_prep2" in n or "k_filter<4, 4, false>" in n) and int(r["Grid_Size"])>100000:
        agg[n[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k, {c: round(sum(x)/len(x)/1e6,2) for c,x in v.items()}, "(millions)")
PY
