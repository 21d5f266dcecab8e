#!/bin/bash
# one-angle-per-wave launches around one workgroup per CU, and pair / triple launches at their limits
for n in 10000 11000 12000 12250 12500 12800 13056; do echo "ncol=$n g=1 $(PICASO_AMD_ANGLE_GROUP=1 python tools/refl_time.py --nwno $n --steps 100 --reps 2 2>&1 | tail -1 | cut -c1-90)"; done
for n in 16000 20000 21000 21700; do echo "ncol=$n g=2 $(PICASO_AMD_ANGLE_GROUP=2 python tools/refl_time.py --nwno $n --steps 100 --reps 2 2>&1 | tail -1 | cut -c1-90)"; done
for n in 25000 30000 32768; do echo "ncol=$n g=3 $(PICASO_AMD_ANGLE_GROUP=3 python tools/refl_time.py --nwno $n --steps 100 --reps 2 2>&1 | tail -1 | cut -c1-90)"; done
