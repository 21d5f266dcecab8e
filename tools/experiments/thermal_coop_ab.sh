#!/bin/bash
# k_thermal_coop helper waves: angles of a layer angle by angle (0) or statement by statement (1)
for v in "-DPZ_COOP_ANGLES_INTERLEAVED=0" "-DPZ_COOP_ANGLES_INTERLEAVED=1"; do
  export PICASO_HIPCC_EXTRA="$v"
  python picaso_amd/build.py --force > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  TAG="$v" python tools/thermal_time.py 10000 16384 2>&1 | grep tag
done
unset PICASO_HIPCC_EXTRA
python picaso_amd/build.py --force > /dev/null 2>&1
