#!/bin/bash
# rocprofv3 recipe (run on the GPU box): kernel trace of the default bench line + separate PMC passes.
# usage: tools/profile.sh <tag> [bench args]   -> gpurun_out/prof_<tag>/..., summaries via collect_profile.py
TAG=${1:-run}; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
# the trace run is the bench line as the driver runs it; PMC passes serialise kernels and need no ramp
CMD="python $ROOT/bench.py --steps 20 --warmup 5 --cpu-sample 0 $*"
PMC="python $ROOT/bench.py --steps 6 --warmup 2 --prewarm-ms 0 --cpu-sample 0 --steady-steps 0 --secondary 0 $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --output-format csv --pmc $pass -d $OUT/pmc_$name -o pmc -- $PMC > $OUT/pmc_$name.log 2>&1
done
cd $ROOT
tail -1 $OUT/trace.log
