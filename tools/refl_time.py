#!/usr/bin/env python
"""Steady-state kernel time of the headline reflected-light launch (run on the GPU box).

The GPU needs a few hundred launches to reach its steady clock state: the first ~20 launches of a
process run ~10 % slower.  This tool pre-warms for --prewarm-ms, then times --steps launches with HIP
events, --reps times, and prints one JSON line per repetition (plus the bit pattern of the result, so
A/B variants can be compared for identity across runs).

    python tools/refl_time.py [--nwno 100000] [--steps 50] [--reps 3] [--ramp] [--batch B] [--lib other.so]

--batch B: B distinct plane sets in ONE launch (picaso_get_reflected_1d_batch_dev), time per SPECTRUM.
--lib: time another build of the library (e.g. the previous round's, kept beside the current one) on the same box.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from picaso_amd import _lib, device, disco, resident  # noqa: E402
from picaso_amd import synthetic as syn  # noqa: E402

TTHG = (1.0, -1.0, 2.0, -0.5, 1.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nwno", type=int, default=100000)
    ap.add_argument("--nlayer", type=int, default=90)
    ap.add_argument("--ng", type=int, default=5)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--prewarm-ms", type=float, default=500.0)
    ap.add_argument("--ramp", action="store_true", help="print the time of consecutive groups of 10 launches from cold")
    ap.add_argument("--tag", default="")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--lib", default="")
    args = ap.parse_args()
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    ctx = _lib.context(0)
    ng, nwno, nlayer = args.ng, args.nwno, args.nlayer
    nlevel = nlayer + 1
    gang, gw, tang, tw = disco.get_angles_1d(ng)
    ubar0, ubar1, _, _, _ = disco.compute_disco(ng, 1, gang, tang, 0.0)
    scene = syn.make_scene(nlayer, nwno, seed=3)
    scene["F0PI"] = np.ones(nwno)
    scene["surf_reflect"] = np.zeros(nwno)
    d = resident.upload_scene(scene, resident.REFLECTED_PLANES + ("F0PI", "surf_reflect"), ctx=ctx)
    xint = device.DeviceArray((ng, 1, nwno), ctx)
    alb = device.DeviceArray((nwno,), ctx)

    B = args.batch
    if B > 1:
        sets = [d]
        for s in range(1, B):
            sc = syn.make_scene(nlayer, nwno, seed=3 + s)
            sc["F0PI"], sc["surf_reflect"] = scene["F0PI"], scene["surf_reflect"]
            sets.append(resident.upload_scene(sc, resident.REFLECTED_PLANES + ("F0PI", "surf_reflect"), ctx=ctx))
        xs = [xint] + [device.DeviceArray((ng, 1, nwno), ctx) for _ in range(B - 1)]
        albs = [alb] + [device.DeviceArray((nwno,), ctx) for _ in range(B - 1)]

    def step():
        if B > 1:
            resident.reflected_1d_batch(ctx, nlevel, nwno, ng, 1, sets, [x["surf_reflect"] for x in sets], ubar0, ubar1,
                                        1.0, [x["F0PI"] for x in sets], 3, 0, *TTHG, xs, gweight=gw, tweight=tw,
                                        albedo=albs)
            return
        resident.reflected_1d(ctx, nlevel, nwno, ng, 1, d, d["surf_reflect"], ubar0, ubar1, 1.0, d["F0PI"], 3, 0,
                              *TTHG, xint, toon_coefficients=0, b_top=0.0, gweight=gw, tweight=tw, albedo=alb)

    def timed(n):
        device.timer_start(ctx)
        for _ in range(n):
            step()
        return device.timer_stop(ctx) / n / B

    abytes = 8 * nwno * (9 * nlayer + 2 * nlevel + 2 + ng + 1)
    if args.ramp:
        device.sync(ctx)
        time.sleep(1.0)
        ms = [timed(10) for _ in range(60)]
        print(json.dumps({"ramp_ms_per_launch_groups_of_10": [round(x, 4) for x in ms]}), flush=True)
    def prewarm(ms):
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < ms:
            for _ in range(20):
                step()
            device.sync(ctx)

    prewarm(args.prewarm_ms)
    out = []
    for r in range(args.reps):
        prewarm(30.0)                       # no idle gap before the timed launches: the clocks drop within ms
        out.append(timed(args.steps))
    x = xint.to_host()
    print(json.dumps({"tag": args.tag, "nwno": nwno, "batch": B, "lib": os.path.basename(_lib.LIB_PATH), "kernel_ms": [round(m, 4) for m in out],
                      "best_hbm_frac": round(abytes / (min(out) * 1e-3) / 8e12, 4),
                      "sha": hashlib.sha1(x.tobytes()).hexdigest()[:12]}), flush=True)


if __name__ == "__main__":
    main()
