#!/bin/bash
# Where do the headline kernel's extra HBM writes come from?  WRITE_SIZE of k_reflected_toa<5, ...> with and without
# the straight-line copy of the delta-scaled cloud-layer body (the one copy that spills registers to scratch).
export TMPDIR=/tmp
ROOT=$PWD
for v in "" "-DPZ_REFL_CLOUD_BODY=0"; do
  export PICASO_HIPCC_EXTRA="$v"
  python picaso_amd/build.py --force > /dev/null 2>&1
  OUT=$ROOT/gpurun_out/wsz_$(echo "$v" | tr -c 'A-Za-z0-9' '_')
  mkdir -p $OUT
  (cd /tmp; rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc -o pmc -- python $ROOT/bench.py --steps 6 --warmup 2 --prewarm-ms 0 --cpu-sample 0 --steady-steps 0 > $OUT/log 2>&1)
  python - <<PY
import csv, glob
vals=[float(r["Counter_Value"]) for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "k_reflected_toa<5" in r["Kernel_Name"] and r["Counter_Name"]=="WRITE_SIZE"]
print("variant [%s]: WRITE_SIZE %.2f KB per launch over %d launches (algorithmic 4687.5 KB)" % ("$v", sum(vals)/max(len(vals),1), len(vals)))
PY
done
unset PICASO_HIPCC_EXTRA
python picaso_amd/build.py --force > /dev/null 2>&1
