#!/bin/bash
# Reflected kernel: time of one spectrum against the number of angles a wave carries (PICASO_AMD_ANGLE_GROUP:
# 0 = all fused, g = groups of g angles as separate waves).  Run on the GPU box; feeds api.hip:reflected_angle_group.
run() { echo "ncol=$1 group=$2 $(PICASO_AMD_ANGLE_GROUP=$2 python tools/refl_time.py --nwno $1 --steps 100 --reps 2 2>&1 | tail -1 | cut -c1-110)"; }
for n in 2000 5000 8000 10000 12500 14000; do for g in 1 2 3; do run $n $g; done; done
for n in 36000 40000 45000 50000 60000 70000; do for g in 0 2 3; do run $n $g; done; done
