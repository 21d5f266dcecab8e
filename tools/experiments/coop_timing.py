#!/usr/bin/env python
"""Per-wave cycle counts of k_reflected_coop's workgroup 0 (build with -DPZ_RCOOP_TIMING)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from picaso_amd import _lib, device, disco, resident
from picaso_amd import synthetic as syn
nwno = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
ctx = _lib.context(0)
ng, nlayer = 5, 90
gang, gw, tang, tw = disco.get_angles_1d(ng)
u0, u1, _, _, _ = disco.compute_disco(ng, 1, gang, tang, 0.0)
sc = syn.make_scene(nlayer, nwno, seed=3)
sc["F0PI"] = np.ones(nwno); sc["surf_reflect"] = np.zeros(nwno)
d = resident.upload_scene(sc, resident.REFLECTED_PLANES + ("F0PI", "surf_reflect"), ctx=ctx)
x = device.DeviceArray((ng, 1, nwno), ctx); alb = device.DeviceArray((nwno,), ctx)
for _ in range(300):
    resident.reflected_1d(ctx, nlayer + 1, nwno, ng, 1, d, d["surf_reflect"], u0, u1, 1.0, d["F0PI"], 3, 0, 1.0, -1.0, 2.0, -0.5, 1.0, x, gweight=gw, tweight=tw, albedo=alb)
device.sync(ctx)
out = (ctypes.c_longlong * 32)()
assert _lib.load().picaso_debug_rcoop(out) == 0
for wv in range(7):
    print("wave %d: total %d cycles, at barriers %d (%.0f %%)" % (wv, out[2 * wv], out[2 * wv + 1], 100.0 * out[2 * wv + 1] / max(out[2 * wv], 1)))
