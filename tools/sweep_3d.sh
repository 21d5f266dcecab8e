#!/bin/bash
# Build kernel variants and time the 3-D facet kernel.  usage: tools/sweep_3d.sh "<flags>" ...
for v in "$@"; do
  export PICASO_HIPCC_EXTRA="$v"
  python picaso_amd/build.py --force > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  echo "=== variant: [$v]"
  BENCH_ONLY=3d python tools/bench_extra.py 2>&1 | grep -A2 "reflected_3d" | tail -2
done
unset PICASO_HIPCC_EXTRA
python picaso_amd/build.py --force > /dev/null 2>&1
