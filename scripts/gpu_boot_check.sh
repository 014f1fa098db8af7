#!/bin/bash
# bootstrap-radius kernel: parity tests, then a kernel trace of the steady-state rebuild (k_boot's average duration)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_regions.py tests/test_config_sizes.py tests/test_device_rebuild.py -m gpu -x -q -k "boot or radius or enlarg or rebuild or region or C3 or c3" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/boot_trace -o st -- python $R/scripts/rebuild_modes.py > $O/boot_trace.log 2>&1
python - $O/boot_trace <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('k_boot', 'k_subtract', 'k_scan', 'k_pack')):
        print(r['Name'][:70].ljust(70), r['Calls'], 'avg us %.1f' % (float(r['AverageNs']) / 1e3), 'min %.1f' % (float(r['MinNs']) / 1e3))
PY
head -30 $O/boot_trace.log
find $O/boot_trace -size +2M -delete
