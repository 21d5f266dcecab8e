#!/bin/bash
for v in "-DPZ_RCOOP_STUB_A -DPZ_RCOOP_STUB_S" "-DPZ_RCOOP_STUB_A" "-DPZ_RCOOP_STUB_S"; do
  export PICASO_HIPCC_EXTRA="$v"
  python picaso_amd/build.py --force > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  echo "=== variant: [$v]"
  python tools/refl_time.py --nwno 12500 --reps 2 --tag="$v" 2>&1 | grep tag
done
unset PICASO_HIPCC_EXTRA
python picaso_amd/build.py --force > /dev/null 2>&1
