#!/usr/bin/env python
"""Steady-state step time of get_thermal_1d + compress_thermal (BASELINE configs[1] and other sizes) --
run on the GPU box.  usage: python tools/thermal_time.py [nwno ...]"""
import hashlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from picaso_amd import _lib, device, disco, resident
from picaso_amd import synthetic as syn
sizes = [int(x) for x in sys.argv[1:]] or [10000]
nlayer, ng = int(os.environ.get("NLAYER", "90")), 5
ctx = _lib.context(0)
gang, gw, tang, tw = disco.get_angles_1d(ng)
_, u1, _, _, _ = disco.compute_disco(ng, 1, gang, tang, 0.0)
for nwno in sizes:
    sc = syn.make_scene(nlayer, nwno, seed=3)
    sc["surf_reflect"] = np.zeros(nwno); sc["dwno"] = sc["wno"] * 0
    d = resident.upload_scene(sc, ("dtau_og", "w0_no_raman", "cosb_og", "wno", "dwno", "surf_reflect"), ctx=ctx)
    f = device.DeviceArray((ng, 1, nwno), ctx); disk = device.DeviceArray((nwno,), ctx)
    def step():
        resident.thermal_1d(ctx, nlayer + 1, d["wno"], nwno, ng, 1, sc["tlevel"], d["dtau_og"], d["w0_no_raman"],
                            d["cosb_og"], sc["plevel"], u1, d["surf_reflect"], 0, f, dwno=d["dwno"], calc_type=0,
                            gweight=gw, tweight=tw, flux_disk=disk)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(20): step()
        device.sync(ctx)
    out = []
    for r in range(3):
        for _ in range(50): step()
        device.timer_start(ctx)
        for _ in range(50): step()
        out.append(device.timer_stop(ctx) / 50)
    ab = 8 * nwno * (3 * nlayer + 3 + ng + 1)
    print(json.dumps({"tag": os.environ.get("TAG", ""), "nwno": nwno, "step_ms": [round(v, 4) for v in out],
                      "hbm_frac": round(ab / (min(out) * 1e-3) / 8e12, 4),
                      "sha": hashlib.sha1(f.to_host().tobytes()).hexdigest()[:10]}))
