#!/bin/bash
# k_reflected_coop: layers per round (one workgroup barrier per round; LDS ring size grows with it)
for v in "-DPZ_RCOOP_ROUND=2" "-DPZ_RCOOP_ROUND=3" "-DPZ_RCOOP_ROUND=4" "-DPZ_RCOOP_ROUND=1" ; do
  export PICASO_HIPCC_EXTRA="$v"
  python picaso_amd/build.py --force > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  echo "=== variant: [$v]"
  for n in 4096 12500 16384; do python tools/refl_time.py --nwno $n --reps 2 --tag="$v" 2>&1 | grep tag; done
done
unset PICASO_HIPCC_EXTRA
python picaso_amd/build.py --force > /dev/null 2>&1
