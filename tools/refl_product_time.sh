#!/bin/bash
# average duration of the reflected kernel inside spectrum(calculation="reflected") (rocprofv3 kernel trace of
# tools/e2e_1d_time.py): the kernel the product runs, alone on the chip; AB_CMD of tools/ab.sh
export TMPDIR=/tmp
R=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d /tmp/rp_XXXX)
(cd /tmp && CALC=reflected WARM=100 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o g -- python $R/tools/e2e_1d_time.py > $D/out.txt 2>&1)
python - "$D" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_reflected_toa" in r["Name"]:
            print("%s avg %.1f us min %.1f x%s" % (r["Name"].split("(")[0][-52:], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Calls"]))
PY
rm -rf $D
