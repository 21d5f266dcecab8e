#!/bin/bash
# A/B of kernel build variants on ONE box: for each flag set rebuild the library and time the headline
# launch in steady state (tools/refl_time.py).   usage: tools/ab.sh "<flags A>" "<flags B>" ...
for v in "$@"; do
  export PICASO_HIPCC_EXTRA="$v"
  python picaso_amd/build.py --force > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  echo "=== variant: [$v]"
  if [ -n "$AB_CMD" ]; then TAG="$v" $AB_CMD 2>&1 | tail -1; else python tools/refl_time.py --reps 3 --tag="$v" ${REFL_ARGS}; fi
done
unset PICASO_HIPCC_EXTRA
python picaso_amd/build.py --force > /dev/null 2>&1
