#!/bin/bash
# k_reflected_coop: which hardware wave (i.e. which SIMD, with which neighbour) runs wave S
for v in "-DPZ_RCOOP_SWAVE=2" "-DPZ_RCOOP_SWAVE=3" "-DPZ_RCOOP_SWAVE=0"; do
  export PICASO_HIPCC_EXTRA="$v"
  python picaso_amd/build.py --force > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  echo "=== variant: [$v]"
  for n in 4096 12500; do python tools/refl_time.py --nwno $n --reps 2 --tag="$v" 2>&1 | grep tag; done
done
unset PICASO_HIPCC_EXTRA
python picaso_amd/build.py --force > /dev/null 2>&1
