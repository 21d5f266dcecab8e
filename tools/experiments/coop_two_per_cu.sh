#!/bin/bash
# k_reflected_coop with rounds of 2 (68 KB of LDS, 123 VGPRs: two workgroups per CU) on 16 385 - 32 768 columns
export PICASO_HIPCC_EXTRA="-DPZ_RCOOP_ROUND=2"
python picaso_amd/build.py --force > /dev/null 2>&1
for n in 12500 20000 25000 32768; do
  PICASO_AMD_REFL_COOP_COLS=32768 python tools/refl_time.py --nwno $n --reps 2 --tag="coop R=2" 2>&1 | grep tag
  PICASO_AMD_REFL_NO_COOP=1 python tools/refl_time.py --nwno $n --reps 2 --tag="grid.y" 2>&1 | grep tag
done
unset PICASO_HIPCC_EXTRA
python picaso_amd/build.py --force > /dev/null 2>&1
