import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from picaso_amd import _lib, device, disco, resident, sharding
from picaso_amd import synthetic as syn
TTHG = (1.0, -1.0, 2.0, -0.5, 1.0)
nlayer, nwno, ng = 40, 30011, 5
nlevel = nlayer + 1
gang, gw, tang, tw = disco.get_angles_1d(ng)
u0, u1, _, _, _ = disco.compute_disco(ng, 1, gang, tang, 0.0)
sc = syn.make_scene(nlayer, nwno, seed=21)
sc["F0PI"] = np.linspace(0.5, 1.5, nwno)
sc["surf_reflect"] = np.full(nwno, 0.2)
keys = resident.REFLECTED_PLANES + ("F0PI", "surf_reflect")
ctx = _lib.context(0)
def run(lo, hi, spread):
    os.environ["PICASO_AMD_SPREAD_COLS"] = str(10**9 if spread else 0)
    d = resident.upload_scene(sc, keys, lo, hi, ctx=ctx)
    x = device.DeviceArray((ng, 1, hi - lo), ctx)
    alb = device.DeviceArray((hi - lo,), ctx)
    resident.reflected_1d(ctx, nlevel, hi - lo, ng, 1, d, d["surf_reflect"], u0, u1, 1.0, d["F0PI"], 3, 0, *TTHG, x,
                          gweight=gw, tweight=tw, albedo=alb)
    device.sync(ctx)
    return x.to_host(), alb.to_host()
xf, af = run(0, nwno, False)
xs, as_ = run(0, nwno, True)
print("fused vs spread, same columns: xint equal", np.array_equal(xf, xs), "ndiff", (xf != xs).sum(), "max rel", np.max(np.abs(xf - xs) / np.abs(xf)),
      "| albedo equal", np.array_equal(af, as_), (af != as_).sum())
for k in range(ng):
    print(" angle", k, "ndiff", (xf[k] != xs[k]).sum())
xh, ah = run(15006, nwno, False)
print("fused shard vs fused full: xint equal", np.array_equal(xh, xf[:, :, 15006:]), (xh != xf[:, :, 15006:]).sum(), "albedo", np.array_equal(ah, af[15006:]))
os.environ["PICASO_AMD_REFL_GENERIC"] = "1"
xg, ag = run(0, nwno, False)
print("generic vs FAST (fused): xint equal", np.array_equal(xg, xf), (xg != xf).sum(), np.max(np.abs(xg - xf) / np.abs(xf)))
