#!/usr/bin/env python
"""All kernels and copies of the last N spectrum() calls of a rocprofv3 trace of tools/e2e_1d_time.py with DEVICES set
(several wavelength blocks): a call = the span between two long idle gaps.  usage: timeline_all.py <dir> [gap_us]"""
import csv, glob, os, sys
d = sys.argv[1]
gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 150e3
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:], "q%s" % r.get("Queue_Id", "")))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[12:], ""))
rows.sort()
calls, cur, last_end = [], [], None
for r in rows:
    if last_end is not None and r[0] - last_end > gap and cur:
        calls.append(cur); cur = []
    cur.append(r)
    last_end = max(last_end or 0, r[1])
if cur:
    calls.append(cur)
c = calls[-2]
t0 = c[0][0]
print("call with %d events, span %.1f us" % (len(c), (max(r[1] for r in c) - t0) / 1e3))
for r in c:
    print("%9.1f %9.1f  %8.1f us  %s %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[2], r[3]))
