#!/usr/bin/env python
"""Steady-state time of the SH4 reflected launch (BASELINE configs[3]) -- run on the GPU box."""
import hashlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from picaso_amd import _lib, device, disco, resident
from picaso_amd import synthetic as syn
TTHG = (1.0, -1.0, 2.0, -0.5, 1.0)
nwno = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nlayer, ng = 90, 5
TOP = int(os.environ.get("TOP", "0"))      # the caller's cloud-free top (picaso_get_reflected_SH_top_dev); the slab starts at 49
ctx = _lib.context(0)
gang, gw, tang, tw = disco.get_angles_1d(ng)
u0, u1, _, _, _ = disco.compute_disco(ng, 1, gang, tang, 0.0)
sc = syn.make_scene(nlayer, nwno, seed=3, stream=4)
sc["F0PI"] = np.ones(nwno); sc["surf_reflect"] = np.zeros(nwno)
d = resident.upload_scene(sc, resident.SH_PLANES + ("F0PI", "surf_reflect"), ctx=ctx)
if os.environ.get("LEAN"):                 # tau / tau_og left out: running products in the kernel (the product's plane set)
    d = {k: v for k, v in d.items() if k not in ("tau", "tau_og")}
x = device.DeviceArray((ng, 1, nwno), ctx); alb = device.DeviceArray((nwno,), ctx)
def step():
    resident.reflected_SH(ctx, nlayer + 1, nwno, ng, 1, d, d["surf_reflect"], u0, u1, 1.0, d["F0PI"], 0, 0, 0, 1, 1, 1,
                          *TTHG, 4, x, gweight=gw, tweight=tw, albedo=alb, cloud_free_above=TOP)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    for _ in range(5): step()
    device.sync(ctx)
out = []
for r in range(3):
    for _ in range(10): step()
    device.timer_start(ctx)
    for _ in range(20): step()
    out.append(device.timer_stop(ctx) / 20)
ab = 8 * nwno * (9 * nlayer + 2 * (nlayer + 1) + 2 + ng + 1)
print(json.dumps({"tag": os.environ.get("TAG", ""), "cloud_free_above": TOP, "lean": bool(os.environ.get("LEAN")), "nwno": nwno, "kernel_ms": [round(v, 4) for v in out],
                  "hbm_frac": round(ab / (min(out) * 1e-3) / 8e12, 4), "sha": hashlib.sha1(x.to_host().tobytes()).hexdigest()[:12]}))
