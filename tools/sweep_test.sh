#!/bin/bash
# Build kernel variants and run a pytest selection for each (run on the GPU box).
# usage: SEL="-k fresh" tools/sweep_test.sh "<flags A>" "<flags B>" ...
for v in "$@"; do
  export PICASO_HIPCC_EXTRA="$v"
  python picaso_amd/build.py --force > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  echo "=== variant: [$v]"
  python -m pytest tests -m gpu -q $SEL 2>&1 | grep -E "^E  +assert|passed|failed" | head -8
done
unset PICASO_HIPCC_EXTRA
python picaso_amd/build.py --force > /dev/null 2>&1
