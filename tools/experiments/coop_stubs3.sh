#!/bin/bash
# k_reflected_coop: is the LDS read burst of the angle waves the hidden bound?  (timing build, wrong results)
for v in "" "-DPZ_RCOOP_STUB_LDS" "-DPZ_RCOOP_STUB_LDS -DPZ_RCOOP_STUB_EXP -DPZ_RCOOP_STUB_RCP"; do
  export PICASO_HIPCC_EXTRA="$v"
  python picaso_amd/build.py --force > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  echo "=== variant: [$v]"
  python tools/refl_time.py --nwno 12500 --reps 2 --tag="$v" 2>&1 | grep tag
done
unset PICASO_HIPCC_EXTRA
python picaso_amd/build.py --force > /dev/null 2>&1
