#!/bin/bash
# Instruction-cache / scalar-cache / wait counters of the headline bench.  usage: tools/pmc_icache.sh <tag>
TAG=${1:-ic}
OUT=$PWD/gpurun_out/pmci_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/bench.py --steps 5 --warmup 1 --cpu-sample 0"
cd /tmp
for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_CYCLES_SMEM" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  n=$(echo $pass | cut -c1-20 | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv --pmc $pass -d $OUT/$n -o pmc -- $CMD > $OUT/$n.log 2>&1
done
cd - > /dev/null
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: [0,0.0])
    for row in csv.DictReader(open(f)):
        if "reflected_toa<5" not in row.get("Kernel_Name",""): continue
        k=row.get("Counter_Name"); agg[k][0]+=1; agg[k][1]+=float(row.get("Counter_Value",0))
    for k,(n,v) in sorted(agg.items()): print("   %-26s mean %.6g" % (k, v/n))
PY
