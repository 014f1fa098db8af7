"""Per-kernel averages of a rocprofv3 --pmc counter_collection csv (big-grid dispatches only)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
mingrid = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    if int(r['Grid_Size']) < mingrid:
        continue
    acc[r['Kernel_Name'][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print('   %-28s %14.0f  (n=%d)' % (c, sum(v) / len(v), len(v)))
