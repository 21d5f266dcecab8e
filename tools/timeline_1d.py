#!/usr/bin/env python
"""Timeline of the last spectrum() calls in a rocprofv3 kernel trace (+ memory-copy trace) of tools/e2e_1d_time.py:
start / end of every kernel and copy relative to the first kernel of the call.
usage: python tools/timeline_1d.py <dir with *_kernel_trace.csv [and *_memory_copy_trace.csv]> [calls_from_end]"""
import csv, glob, os, sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:], "q%s" % r.get("Queue_Id", "")))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", ""), ""))
rows.sort()
# a call starts at every k_opacity_gas kernel
starts = [i for i, r in enumerate(rows) if "k_opacity_gas" in r[2]]
for s in starts[-back - 1:-1]:
    e = next((j for j in starts if j > s), len(rows))
    t0 = rows[s][0]
    print("--- call (next call's first kernel %.1f us after this one's)" % ((rows[e][0] - t0) / 1e3 if e < len(rows) else -1))
    for r in rows[s:e]:
        print("%9.1f %9.1f  %8.1f us  %s %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[2], r[3]))
