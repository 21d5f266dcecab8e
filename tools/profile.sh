#!/bin/bash
# rocprofv3 recipe (run on the GPU box): kernel trace + separate PMC passes for the headline bench.
# usage: tools/profile.sh <tag>      -> gpurun_out/prof_<tag>/...
TAG=${1:-run}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/bench.py --steps 10 --warmup 2 --cpu-sample 0"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --output-format csv --pmc $pass -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1
done
cd - > /dev/null
python - <<PY
import csv, glob, os, collections
out="$OUT"
for f in sorted(glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True)):
    print("== kernel stats", f.replace(out,""))
    for row in list(csv.reader(open(f)))[:6]: print("   ", row)
for f in sorted(glob.glob(out+"/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: [0,0.0])
    for row in csv.DictReader(open(f)):
        k=(row.get("Kernel_Name","")[:40], row.get("Counter_Name"))
        agg[k][0]+=1; agg[k][1]+=float(row.get("Counter_Value",0))
    print("== pmc", f.replace(out,""))
    for k,(n,v) in sorted(agg.items()):
        if "reflected" in k[0] or "thermal" in k[0]: print("    %-42s %-28s dispatches %d  mean %.6g" % (k[0],k[1],n,v/n))
PY
