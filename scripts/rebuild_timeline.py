#!/usr/bin/env python3
"""Merged timeline (kernels + memory copies) of the LAST rebuild in a rocprofv3 trace of scripts/rebuild_trace.py.
    python scripts/rebuild_timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv>"""
import csv
import glob
import os
import sys

d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-46:]))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s B" % (r.get("Direction", "?"), r.get("Size", r.get("Bytes", "?")))))
ev.sort()
# the last rebuild: from the last k_subtract_accum backwards to the copy in front of it
starts = [i for i, e in enumerate(ev) if "subtract" in e[2]]
i0 = starts[-1]
while i0 > 0 and ev[i0][0] - ev[i0 - 1][1] < 150000 and "COPY" in ev[i0 - 1][2]:
    i0 -= 1
t0 = ev[i0][0]
prev_end = t0
busy = 0
for s, e, name in ev[i0:]:
    print("%9.1f us  +%7.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name))
    busy += e - s
    prev_end = max(prev_end, e)
print("span %.1f us, device busy %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3))
