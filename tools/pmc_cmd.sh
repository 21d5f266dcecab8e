#!/bin/bash
# One PMC pass (instruction counts) around an arbitrary command.  usage: tools/pmc_cmd.sh <tag> <cmd...>
TAG=$1; shift
OUT=$PWD/gpurun_out/pmcc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc -o pmc -- "$@" > $OUT/pmc.log 2>&1
cd $ROOT
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: [0,0.0])
    for row in csv.DictReader(open(f)):
        k=(row.get("Kernel_Name","")[:52], row.get("Counter_Name"))
        agg[k][0]+=1; agg[k][1]+=float(row.get("Counter_Value",0))
    for k,(n,v) in sorted(agg.items()):
        if k[1] in ("SQ_INSTS_VALU","SQ_WAVES","SQ_INSTS_SALU"): print("    %-54s %-16s n %d  mean %.6g" % (k[0],k[1],n,v/n))
PY
