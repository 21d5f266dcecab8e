#!/usr/bin/env python
"""Steady-state time of get_thermal_SH (SH4, 1e5 x 90 x 5; k_sh_thermal) -- run on the GPU box."""
import hashlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from picaso_amd import _lib, device, disco, resident
from picaso_amd.spectrum import _thermal_sh
from picaso_amd import synthetic as syn
nwno = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nlayer, ng = 90, 5
ctx = _lib.context(0)
gang, gw, tang, tw = disco.get_angles_1d(ng)
_, u1, _, _, _ = disco.compute_disco(ng, 1, gang, tang, 0.0)
sc = syn.make_scene(nlayer, nwno, seed=3, stream=4)
sc["surf_reflect"] = np.zeros(nwno)
d = resident.upload_scene(sc, ("dtau", "w0", "cosb_og", "wno", "surf_reflect"), ctx=ctx)
f = device.DeviceArray((ng, 1, nwno), ctx); disk = device.DeviceArray((nwno,), ctx)
def step():
    _thermal_sh(ctx, nlayer + 1, d["wno"], nwno, ng, 1, sc["tlevel"], d, sc["plevel"], u1, d["surf_reflect"], 4, 0, True, f,
                gw, tw, disk)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    for _ in range(10): step()
    device.sync(ctx)
out = []
for _ in range(3):
    device.timer_start(ctx)
    for _ in range(30): step()
    out.append(round(device.timer_stop(ctx) / 30, 4))
print(json.dumps({"nwno": nwno, "step_ms": out, "sha": hashlib.sha1(disk.to_host().tobytes()).hexdigest()[:10]}))
