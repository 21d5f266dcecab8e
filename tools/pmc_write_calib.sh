#!/bin/bash
# Calibrate rocprofv3's WRITE_SIZE / FETCH_SIZE on this box against a kernel with a known byte count
# (k_broadcast_facets: reads rows*nwno doubles once, writes rows*nwno*64 doubles, 8 B per lane, fully coalesced).
# MI355X_MICROARCH.md: FETCH_SIZE reports half of a wide streaming read on gfx950, WRITE_SIZE is uncalibrated.
# usage: tools/pmc_write_calib.sh <tag>    -> gpurun_out/calib_<tag>/calib.json
TAG=${1:-c}
OUT=$PWD/gpurun_out/calib_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cat > $OUT/run.py <<PY
import sys
sys.path.insert(0, "$ROOT")
import numpy as np
from picaso_amd import _lib, device
ctx = _lib.context(0)
src = device.DeviceArray.from_host(np.random.default_rng(0).random((100, 10000)), ctx)
for _ in range(5):
    out = device.broadcast_facets(src, 64, None, ctx)
    device.sync(ctx)
    out.free()
PY
cd /tmp
for pass in "WRITE_SIZE" "FETCH_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --output-format csv --pmc $pass -d $OUT/pmc_$name -o pmc -- python $OUT/run.py > $OUT/pmc_$name.log 2>&1
done
cd $ROOT
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: [0, 0.0])
for f in sorted(glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "k_broadcast_facets" in row.get("Kernel_Name", ""):
            k = row.get("Counter_Name")
            agg[k][0] += 1
            agg[k][1] += float(row.get("Counter_Value", 0))
m = {k: v / n for k, (n, v) in agg.items()}
written, read = 100 * 10000 * 64 * 8, 100 * 10000 * 8
res = {"kernel": "k_broadcast_facets (100 x 10000 -> x 64 facets)", "bytes_written": written, "bytes_read": read,
       "counters_per_dispatch": m}
if "WRITE_SIZE" in m:
    res["WRITE_SIZE_KB_over_written_KB"] = m["WRITE_SIZE"] * 1024.0 / written
if "FETCH_SIZE" in m:
    res["FETCH_SIZE_KB_over_read_KB"] = m["FETCH_SIZE"] * 1024.0 / read
json.dump(res, open("$OUT/calib.json", "w"), indent=1)
print(json.dumps(res))
PY
