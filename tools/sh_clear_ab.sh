#!/bin/bash
# The measured A/B of the SH4 angle sharing (review item 7): run on the GPU box, writes gpurun_out/sh4_clear_ab.json
# (copied to profiles/r04_sh4_clear_ab.json) -- kernel times of the full-plane launch and of the cloud-free form at one
# and two angles per lane, at 1e5 / 3e4 / 12 500 columns, the instruction counts of both kernels (one PMC pass), and the
# product call spectrum(rt_method='SH', stream=4) with and without the cloud-free form; cloud_deck_scene: BASELINE configs[3]'s
# scene (cloud slab in layers 49-58 of 90) without and with the caller's statement cloud_free_above = 49
# (picaso_get_reflected_SH_top_dev).
mkdir -p gpurun_out
{
  echo '{"kernel": ['
  PICASO_AMD_SHC_ANGLES=1 python tools/sh_clear_time.py 100000; echo ,
  PICASO_AMD_SHC_ANGLES=2 python tools/sh_clear_time.py 100000; echo ,
  python tools/sh_clear_time.py 30000; echo ,
  python tools/sh_clear_time.py 12500
  echo '], "cloud_deck_scene": ['
  TOP=0 python tools/sh_time.py 100000; echo ,
  TOP=49 python tools/sh_time.py 100000; echo ,
  TOP=0 python tools/sh_time.py 12500; echo ,
  TOP=49 python tools/sh_time.py 12500
  echo '], "product_ms": {'
  echo '"sh4_reflected_thermal_clear":' $(RT=SH python tools/e2e_1d_time.py | python -c "import json,sys; print(list(json.loads(sys.stdin.read()).values())[0])") ,
  echo '"sh4_reflected_thermal_all_planes":' $(RT=SH PICASO_AMD_ALL_PLANES=1 python tools/e2e_1d_time.py | python -c "import json,sys; print(list(json.loads(sys.stdin.read()).values())[0])") ,
  echo '"sh4_reflected_clear":' $(RT=SH CALC=reflected python tools/e2e_1d_time.py | python -c "import json,sys; print(list(json.loads(sys.stdin.read()).values())[0])") ,
  echo '"sh4_reflected_all_planes":' $(RT=SH CALC=reflected PICASO_AMD_ALL_PLANES=1 python tools/e2e_1d_time.py | python -c "import json,sys; print(list(json.loads(sys.stdin.read()).values())[0])")
  echo '}, "pmc": "'
  bash tools/pmc_cmd.sh shclear python $PWD/tools/sh_clear_time.py 100000 | grep -E "k_sh4_clear|k_sh<" | sed 's/"/ /g'
  echo '"}'
} > gpurun_out/sh4_clear_ab.txt
python - <<'PY'
import json, re
t = open("gpurun_out/sh4_clear_ab.txt").read()
head, pmc = t.split('"pmc": "')
pmc_lines = [" ".join(l.split()) for l in pmc.strip().rstrip('"}').strip().splitlines() if l.strip()]
d = json.loads(head + '"pmc": []}')
d["pmc"] = pmc_lines
json.dump(d, open("gpurun_out/sh4_clear_ab.json", "w"), indent=1)
print(json.dumps(d, indent=1))
PY
