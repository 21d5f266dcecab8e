"""Per-kernel register / scratch / LDS table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/resource_usage.py picaso_amd/csrc/toon_reflected.hip [extra hipcc flags] > before.txt

Used to check that a change meant for one launch shape leaves the other instantiations' register
allocation alone (diff the tables of two builds).
"""
import re
import subprocess
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from picaso_amd import build as b


def table(src, extra=()):
    cmd = [b.HIPCC] + b.FLAGS + list(extra) + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: .*?(Function Name|Name): (\S+)", line)
        if m:
            cur = {"name": m.group(2)}
            rows.append(cur)
            continue
        m = re.search(r"remark: .*?\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return rows


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE,
                       text=True)
    return p.stdout.splitlines()


if __name__ == "__main__":
    rows = table(sys.argv[1], sys.argv[2:])
    names = demangle([r["name"] for r in rows])
    for r, nm in zip(rows, names):
        nm = re.sub(r"\(.*\)$", "", nm).replace("void pz::", "")
        print("%-62s vgpr %3d sgpr %3d scratch %4d vspill %3d sspill %3d lds %6d occ %d" % (
            nm, r.get("VGPRs", -1), r.get("TotalSGPRs", -1), r.get("ScratchSize", -1), r.get("VGPRs Spill", -1),
            r.get("SGPRs Spill", -1), r.get("LDS Size", -1), r.get("Occupancy", -1)))
