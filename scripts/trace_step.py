"""Print the kernels of one headline step (P = 10^6) from a rocprofv3 kernel trace csv: start, duration, gap."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
key = sys.argv[2] if len(sys.argv) > 2 else 'k_prep4'
idx = [i for i, r in enumerate(rows) if key in r['Kernel_Name'] and int(r['Grid_Size_X']) > 100000]
i0, i1 = idx[-3], idx[-2]
prev_end = None
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1 + 1]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (st - prev_end) if prev_end else 0
    print(r['Kernel_Name'][:56].ljust(56), 'start %7.1f dur %7.1f us  gap %5.1f us' % ((st - t0) / 1e3, (en - st) / 1e3, gap / 1e3),
          r['Grid_Size_X'], 'vgpr', r['VGPR_Count'])
    prev_end = en
