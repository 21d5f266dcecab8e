#!/bin/bash
# rocprofv3 kernel stats + HBM / VALU counters of the secondary BASELINE workloads (bench.py --config 1|3|4),
# run on the GPU box.  usage: tools/profile_configs.sh <tag>   -> gpurun_out/prof_<tag>_c<N>/ and profiles/
TAG=${1:-r02}
export TMPDIR=/tmp
ROOT=$PWD
for C in 1 3 4 shard; do
  OUT=$ROOT/gpurun_out/prof_${TAG}_c$C
  mkdir -p $OUT
  ARGS="--config $C"
  case $C in 1) K="k_thermal_";; 3) K="k_sh_refl<";; 4) K="k_reflected_toa<1, true";;
             shard) K="k_reflected_coop"; ARGS="--config 2 --nwno 12500";; esac     # the 8-GPU wavelength shard
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/bench.py $ARGS --steps 20 --warmup 5 --cpu-sample 0 > $OUT/trace.log 2>&1
  for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
    name=$(echo $pass | tr ' ' '_' | cut -c1-40)
    rocprofv3 --kernel-trace --output-format csv --pmc $pass -d $OUT/pmc_$name -o pmc -- python $ROOT/bench.py $ARGS --steps 6 --warmup 2 --prewarm-ms 0 --cpu-sample 0 --steady-steps 0 > $OUT/pmc_$name.log 2>&1
  done
  cd $ROOT
  python tools/collect_profile.py $OUT ${TAG}_config$C "$K" > $OUT/summary.txt 2>&1
  cp profiles/${TAG}_config${C}_* $OUT/ 2>/dev/null
  tail -1 $OUT/trace.log | cut -c1-400
done
