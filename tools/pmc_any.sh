#!/bin/bash
# PMC passes around an arbitrary command; prints per-kernel means.  usage: tools/pmc_any.sh <tag> "<counters pass 1>|<pass 2>|..." <cmd...>
TAG=$1; PASSES=$2; shift 2
OUT=$PWD/gpurun_out/pmca_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
IFS='|' read -ra PS <<< "$PASSES"
i=0
for pass in "${PS[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $pass -d $OUT/pmc_$i -o pmc -- "$@" > $OUT/pmc_$i.log 2>&1
done
cd $ROOT
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: [0,0.0])
for f in sorted(glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k=(row.get("Kernel_Name","")[:60], row.get("Counter_Name"))
        agg[k][0]+=1; agg[k][1]+=float(row.get("Counter_Value",0))
for k,(n,v) in sorted(agg.items()):
    print("    %-62s %-22s n %d  mean %.6g" % (k[0],k[1],n,v/n))
PY
