#!/usr/bin/env python
"""SH4 reflected light on a cloud-free 1e5 x 90 x 5 spectrum: the full-plane launch (k_sh, eleven planes) against the
cloud-free form (k_sh4_clear: dtau and w0 only, angle-independent half shared between the angles of a lane) -- run on
the GPU box.  PICASO_AMD_SHC_ANGLES=1..5 fixes the angles per lane, PICASO_AMD_SHC_ONE_WAVE=1 the one-wave-per-SIMD
build (all five angles in registers without scratch)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from picaso_amd import _lib, device, disco, resident
from picaso_amd import synthetic as syn
TTHG = (1.0, -1.0, 2.0, -0.5, 1.0)
nwno = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nlayer, ng = 90, 5
ctx = _lib.context(0)
gang, gw, tang, tw = disco.get_angles_1d(ng)
u0, u1, _, _, _ = disco.compute_disco(ng, 1, gang, tang, 0.0)
sc = syn.make_scene(nlayer, nwno, seed=3, stream=4, cloud=False)
sc["F0PI"] = np.ones(nwno); sc["surf_reflect"] = np.zeros(nwno)
d = resident.upload_scene(sc, resident.SH_PLANES + ("F0PI", "surf_reflect"), ctx=ctx)
lean = {"dtau": d["dtau"], "w0": d["w0"]}
x = device.DeviceArray((ng, 1, nwno), ctx); alb = device.DeviceArray((nwno,), ctx)


def step(planes):
    resident.reflected_SH(ctx, nlayer + 1, nwno, ng, 1, planes, d["surf_reflect"], u0, u1, 1.0, d["F0PI"], 0, 0, 0, 1, 1, 1,
                          *TTHG, 4, x, gweight=gw, tweight=tw, albedo=alb)


def timed(planes):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        for _ in range(5): step(planes)
        device.sync(ctx)
    out = []
    for r in range(3):
        for _ in range(10): step(planes)
        device.timer_start(ctx)
        for _ in range(20): step(planes)
        out.append(device.timer_stop(ctx) / 20)
    return [round(v, 4) for v in out], x.to_host()


res = {"tag": os.environ.get("TAG", ""), "nwno": nwno, "angles_per_lane": os.environ.get("PICASO_AMD_SHC_ANGLES", "model"),
       "one_wave": bool(os.environ.get("PICASO_AMD_SHC_ONE_WAVE"))}
if not os.environ.get("CLEAR_ONLY"):
    res["full_ms"], xf = timed(d)
res["clear_ms"], xc = timed(lean)
if "full_ms" in res:
    res["max_rel_diff"] = float(np.max(np.abs(xc - xf) / np.abs(xf)))
    res["ratio"] = round(min(res["clear_ms"]) / min(res["full_ms"]), 4)
print(json.dumps(res))
