#!/bin/bash
# Build kernel variants; for each, time the headline kernel single-stream and with 2 streams, in one call
# (same box).  usage: tools/sweep_pipe.sh "<flags A>" "<flags B>" ...
for v in "$@"; do
  export PICASO_HIPCC_EXTRA="$v"
  python picaso_amd/build.py --force > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  echo "=== variant: [$v]"
  BENCH_ONLY=pipeline python tools/bench_extra.py | grep -A1 "1stream\|2stream" | grep ms_per
done
unset PICASO_HIPCC_EXTRA
python picaso_amd/build.py --force > /dev/null 2>&1
