#!/bin/bash
# activity counters of the rebuild kernels (one rocprofv3 --pmc pass over scripts/rebuild_modes.py); prints k_boot's averages
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/boot_pmc -o st -- python $R/scripts/rebuild_modes.py > $O/boot_pmc.log 2>&1
python - $O/boot_pmc <<'PY'
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'k_boot<' in n or 'k_scan<50, 16, false' in n or 'k_subtract' in n:
        acc[n[:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for n, d in acc.items():
    print(n, {k: round(sum(v) / len(v)) for k, v in d.items()})
PY
find $O/boot_pmc -size +2M -delete
