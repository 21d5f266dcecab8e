#!/usr/bin/env python
"""Condense a tools/profile.sh output directory (gpurun_out/prof_<tag>) into the small summaries kept
under profiles/: <name>_kernel_stats.csv, <name>_pmc_summary.csv and traffic.json.

    python tools/collect_profile.py gpurun_out/prof_r01v3 r01_v3
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, name = sys.argv[1], sys.argv[2]
kernel_tag = sys.argv[3] if len(sys.argv) > 3 else "k_reflected_toa<5"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, name + "_kernel_stats.csv"))
# the same trace split by launch shape: bench.py's companions run one kernel at several sizes (k_sh at 1e5 and at
# 12 500 columns, the batched launch at B = 4 and 8), which the per-kernel stats average together
trace = glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True)
if trace:
    by = collections.defaultdict(list)
    for row in csv.DictReader(open(trace[0])):
        grid = "x".join(row[k] for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        by[(row["Kernel_Name"], grid)].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    with open(os.path.join(dst, name + "_kernel_stats_by_grid.csv"), "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "grid_threads", "calls", "average_ns", "median_ns", "min_ns", "max_ns"])
        for (kern, grid), d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
            if "pz::" not in kern:
                continue
            d = sorted(d)
            w.writerow([kern, grid, len(d), "%.0f" % (sum(d) / len(d)), d[len(d) // 2], d[0], d[-1]])
agg = collections.defaultdict(lambda: [0, 0.0])
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = (row.get("Kernel_Name", ""), row.get("Counter_Name"))
        agg[k][0] += 1
        agg[k][1] += float(row.get("Counter_Value", 0))
with open(os.path.join(dst, name + "_pmc_summary.csv"), "w") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
    for (kern, ctr), (n, v) in sorted(agg.items()):
        w.writerow([kern, ctr, n, "%.6g" % (v / n)])
# HBM traffic of the headline kernel: FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE under-counts by
# 2x on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section), collected in separate passes.
fetch = [v / n for (k, c), (n, v) in agg.items() if c == "FETCH_SIZE" and kernel_tag in k]
write = [v / n for (k, c), (n, v) in agg.items() if c == "WRITE_SIZE" and kernel_tag in k]
valu = [v / n for (k, c), (n, v) in agg.items() if c == "SQ_INSTS_VALU" and kernel_tag in k]
if fetch and write and kernel_tag == "k_reflected_toa<5":
    sys.path.insert(0, root)
    import bench
    json.dump({"hbm_bytes_per_launch": (2.0 * fetch[0] + write[0]) * 1024.0,
               "valu_wave_insts_per_launch": valu[0] if valu else None,
               "kernel_source_hash": bench.kernel_source_hash(),
               "source": "profiles/%s_pmc_summary.csv" % name,
               "note": "(2*FETCH_SIZE + WRITE_SIZE) KB per dispatch of the headline k_reflected_toa launch; "
                       "FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md; bench.py only "
                       "quotes these numbers while kernel_source_hash matches the sources it runs"},
              open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, name + "_pmc_summary.csv")).read())
