#!/bin/bash
# Build kernel variants (-D macros via PICASO_HIPCC_EXTRA) and time the headline bench for each.
# usage: tools/sweep.sh "<flags A>" "<flags B>" ...   (run on the GPU box)
for v in "$@"; do
  export PICASO_HIPCC_EXTRA="$v"
  python picaso_amd/build.py --force > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  echo "=== variant: [$v]"
  python bench.py --steps 20 --warmup 3 --cpu-sample ${CPU_SAMPLE:-2000} | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('  value %.1f spectra/s  kernel_ms %.4f  frac %.3f  err %s' % (d['value'], r['kernel_ms'], r['frac'], d.get('max_rel_err_vs_oracle')))"
done
unset PICASO_HIPCC_EXTRA
python picaso_amd/build.py --force > /dev/null 2>&1
