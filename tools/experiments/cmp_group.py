import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from picaso_amd import _lib, device, disco, resident
from picaso_amd import synthetic as syn
TTHG = (1.0, -1.0, 2.0, -0.5, 1.0)
ctx = _lib.context(0)
ng, nlayer = 5, 90
for nwno, sr in ((12500, 0.0), (12500, 0.1), (25000, 0.0)):
    gang, gw, tang, tw = disco.get_angles_1d(ng)
    ubar0, ubar1, _, _, _ = disco.compute_disco(ng, 1, gang, tang, 0.0)
    scene = syn.make_scene(nlayer, nwno, seed=3)
    scene["F0PI"] = np.ones(nwno); scene["surf_reflect"] = np.full(nwno, sr)
    d = resident.upload_scene(scene, resident.REFLECTED_PLANES + ("F0PI", "surf_reflect"), ctx=ctx)
    res = {}
    for g in (0, 1, 2, 3):
        os.environ["PICASO_AMD_ANGLE_GROUP"] = str(g)
        xint = device.DeviceArray.zeros((ng, 1, nwno), ctx); alb = device.DeviceArray.zeros((nwno,), ctx)
        resident.reflected_1d(ctx, nlayer + 1, nwno, ng, 1, d, d["surf_reflect"], ubar0, ubar1, 1.0, d["F0PI"], 3, 0,
                              *TTHG, xint, toon_coefficients=0, b_top=0.0, gweight=gw, tweight=tw, albedo=alb)
        res[g] = (xint.to_host(), alb.to_host())
    for g in (1, 2, 3):
        dx = res[g][0] != res[0][0]
        print(nwno, sr, "group", g, "x differs at", int(dx.sum()), "alb differs at", int((res[g][1] != res[0][1]).sum()),
              "max rel", float(np.max(np.abs(res[g][0] - res[0][0]) / np.abs(res[0][0]))))
        if dx.sum():
            a, _, c = np.nonzero(dx)
            print("   angles", np.unique(a), "cols", c[:10], "neg zero?", np.signbit(res[g][0][dx][:5]), res[0][0][dx][:5], res[g][0][dx][:5])
