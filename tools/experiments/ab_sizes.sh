#!/bin/bash
# A/B of build flags over the shard sizes: usage tools/experiments/ab_sizes.sh "<flags A>" "<flags B>" ...
for v in "$@"; do
  export PICASO_HIPCC_EXTRA="$v"
  python picaso_amd/build.py --force > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  echo "=== variant: [$v]"
  for n in 12500 25000 50000 100000 200000; do python tools/refl_time.py --nwno $n --steps 100 --reps 3 | tail -1 | cut -c1-110; done
done
unset PICASO_HIPCC_EXTRA
python picaso_amd/build.py --force > /dev/null 2>&1
