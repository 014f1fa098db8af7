#!/bin/bash
# kernel trace of device-resident mid-size calls: scripts/gpu_trace_mid.sh <tag> <batch> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for B in "$@"; do
rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_trace_$B -o st -- python $R/scripts/midsize_profile.py $B > $O/${TAG}_trace_$B.log 2>&1
python - $O/${TAG}_trace_$B <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_prep4' in r['Kernel_Name']]
i0, i1 = idx[-3], idx[-2]
prev = None
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1 + 1]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(r['Kernel_Name'][:60].ljust(60), 'start %7.1f dur %6.1f gap %5.1f us' % ((st - t0) / 1e3, (en - st) / 1e3, ((st - prev) / 1e3) if prev else 0), 'grid', r['Grid_Size_X'], r.get('Grid_Size_Y', ''), 'vgpr', r.get('VGPR_Count', ''))
    prev = en
PY
tail -1 $O/${TAG}_trace_$B.log
find $O/${TAG}_trace_$B -size +2M -delete
done
