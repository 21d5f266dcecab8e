#!/bin/bash
# average duration of the opacity kernels inside spectrum() (rocprofv3 kernel trace of tools/e2e_1d_time.py); used as
# AB_CMD of tools/ab.sh (GAS_TOOL=tools/e2e_3d_time.py: the 3-D path's launches)
export TMPDIR=/tmp
R=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d /tmp/gas_XXXX)
(cd /tmp && WARM=80 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o g -- python $R/${GAS_TOOL:-tools/e2e_1d_time.py} > $D/out.txt 2>&1)
python - "$D" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
out = []
for r in rows:
    n = r["Name"]
    if "k_opacity_gas" in n or "k_level_sums" in n or "k_compute_opacity" in n:
        out.append("%s %.1f us x%s" % (n.split("(")[0][-22:], float(r["AverageNs"]) / 1e3, r["Calls"]))
print("; ".join(out), "|", open(sys.argv[1] + "/out.txt").read().strip().splitlines()[-1][:80])
PY
rm -rf $D
