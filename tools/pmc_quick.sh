#!/bin/bash
# One PMC pass (instruction counts) of the headline bench.  usage: tools/pmc_quick.sh <tag> [bench args]
TAG=${1:-q}; shift
OUT=$PWD/gpurun_out/pmcq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/bench.py --steps 5 --warmup 1 --cpu-sample 0 $@"
cd /tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc -o pmc -- $CMD > $OUT/pmc.log 2>&1
cd - > /dev/null
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: [0,0.0])
    for row in csv.DictReader(open(f)):
        k=(row.get("Kernel_Name","")[:44], row.get("Counter_Name"))
        agg[k][0]+=1; agg[k][1]+=float(row.get("Counter_Value",0))
    for k,(n,v) in sorted(agg.items()):
        print("    %-46s %-18s n %d  mean %.6g" % (k[0],k[1],n,v/n))
PY
