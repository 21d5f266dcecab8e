#!/bin/bash
# fused five-angle launch: sweep state in registers (one wave per SIMD, default up to 1 024 column-waves) against state in LDS (PICASO_AMD_REFL_NO_BIG)
for n in 40000 50000 60000 65536; do
  echo "ncol=$n lds     $(PICASO_AMD_ANGLE_GROUP=0 PICASO_AMD_REFL_NO_BIG=1 python tools/refl_time.py --nwno $n --steps 100 --reps 2 | tail -1 | cut -c1-100)"
  echo "ncol=$n big     $(PICASO_AMD_ANGLE_GROUP=0 python tools/refl_time.py --nwno $n --steps 100 --reps 2 | tail -1 | cut -c1-100)"
done
