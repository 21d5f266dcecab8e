#!/usr/bin/env python
"""Regenerate the table of numbers in profiles/README.md FROM the committed CSVs, so the text cannot drift from
the files it describes: for every <name>_kernel_stats.csv / <name>_pmc_summary.csv pair of a round, the dominant
kernel's calls / average / min / max and its counters per dispatch.

    python tools/profiles_readme.py r03          # rewrites the block between the GENERATED markers
"""
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
COUNTERS = ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "FETCH_SIZE", "WRITE_SIZE")
ALGORITHMIC = {   # algorithmic HBM bytes of one launch of the named bench workloads
    "headline": 8 * 100000 * (9 * 90 + 2 * 91 + 2 + 5 + 1),
    "configshard": 8 * 12500 * (9 * 90 + 2 * 91 + 2 + 5 + 1),
    "config1": 8 * 10000 * (3 * 90 + 3 + 5 + 1),
    "config3": 8 * 100000 * (9 * 90 + 2 * 91 + 2 + 5 + 1),
    "config4": 8 * 12500 * (64 * (9 * 90 + 2 * 91 + 1) + 2 + 1),
}


def short(name):
    m = re.search(r"pz::(k_[A-Za-z0-9_]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def main():
    rnd = sys.argv[1]
    rows = []
    for f in sorted(glob.glob(os.path.join(PROF, rnd + "_*_kernel_stats.csv"))):
        tag = os.path.basename(f)[len(rnd) + 1:-len("_kernel_stats.csv")]
        with open(f) as fh:
            stats = sorted(csv.DictReader(fh), key=lambda r: -float(r["TotalDurationNs"]))
        if not stats:
            continue
        top = stats[0]
        pmc = {}
        pf = os.path.join(PROF, "%s_%s_pmc_summary.csv" % (rnd, tag))
        if os.path.exists(pf):
            with open(pf) as fh:
                for r in csv.DictReader(fh):
                    if r["kernel"] == top["Name"]:
                        pmc[r["counter"]] = float(r["mean_per_dispatch"])
        cell = ", ".join("`%s` %.6g" % (c, pmc[c]) for c in COUNTERS if c in pmc)
        traffic = ""
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            t = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
            traffic = "%.1f MB" % (t / 1e6)
            if tag in ALGORITHMIC:
                traffic += " = %.3f x algorithmic (%.1f MB)" % (t / ALGORITHMIC[tag], ALGORITHMIC[tag] / 1e6)
        avg = float(top["AverageNs"]) / 1e6
        bw = ""
        if tag in ALGORITHMIC:
            bw = "%.2f TB/s = %.3f of 8 TB/s" % (ALGORITHMIC[tag] / (avg * 1e-3) / 1e12,
                                                  ALGORITHMIC[tag] / (avg * 1e-3) / 8e12)
        rows.append("| `%s_%s_*` | `%s` | %s | %.4f | %.4f | %.4f | %s | %s | %s |" % (
            rnd, tag, short(top["Name"]), top["Calls"], avg, float(top["MinNs"]) / 1e6, float(top["MaxNs"]) / 1e6,
            bw, cell or "-", traffic or "-"))
    block = ["<!-- GENERATED %s: tools/profiles_readme.py %s -->" % (rnd, rnd),
             "| files | dominant kernel | calls | avg ms | min ms | max ms | algorithmic bytes / avg | counters per dispatch "
             "(KB for FETCH_SIZE / WRITE_SIZE) | 2 x FETCH_SIZE + WRITE_SIZE |",
             "|---|---|---|---|---|---|---|---|---|"] + rows + ["<!-- END GENERATED %s -->" % rnd]
    path = os.path.join(PROF, "README.md")
    text = open(path).read()
    pat = re.compile(r"<!-- GENERATED %s:.*?<!-- END GENERATED %s -->" % (rnd, rnd), re.S)
    new = "\n".join(block)
    if pat.search(text):
        text = pat.sub(lambda m: new, text)
    else:
        text = text.replace("# rocprofv3 summaries\n", "# rocprofv3 summaries\n\n## Round %s (numbers generated from "
                            "the CSVs)\n\n%s\n" % (rnd[1:].lstrip("0"), new), 1)
    open(path, "w").write(text)
    print(new)


if __name__ == "__main__":
    main()
