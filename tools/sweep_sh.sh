#!/bin/bash
# Build kernel variants and time the SH4 reflected spectrum.  usage: tools/sweep_sh.sh "<flags>" ...
for v in "$@"; do
  export PICASO_HIPCC_EXTRA="$v"
  python picaso_amd/build.py --force > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  echo "=== variant: [$v]"
  BENCH_ONLY=sh python tools/bench_extra.py 2>&1 | grep -A1 "SH4_100000" | tail -1
done
unset PICASO_HIPCC_EXTRA
python picaso_amd/build.py --force > /dev/null 2>&1
