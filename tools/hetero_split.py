#!/usr/bin/env python
"""Experiment (round 3): heterogeneous angle split of the 1e5-column headline launch.

The single fused launch is 391 workgroups on 256 CUs x 2 slots: 135 CUs carry two five-angle sweeps, 121 one.
Here the first N1 column blocks run the fused five-angle kernel on one stream and the remaining columns run as
angle GROUPS (PICASO_AMD_ANGLE_GROUP=g: ceil(5/g) workgroups per column block, each carrying g angles) on a second
stream at the same time, so that the chip holds ~512 workgroups of unequal size.  Prints ms per spectrum for a
sweep of N1 (in 256-column blocks) and g.

    python tools/hetero_split.py [--steps 200]
"""
import argparse
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
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--blocks", default="391,320,300,285,270,256,240,220")
    ap.add_argument("--groups", default="3,2")
    args = ap.parse_args()
    ctx = _lib.context(0)
    aux = _lib.new_context(0)
    ng, nwno, nlayer = 5, args.nwno, 90
    nlevel = nlayer + 1
    gang, gw, tang, tw = disco.get_angles_1d(ng)
    ubar0, ubar1, _, _, _ = disco.compute_disco(ng, 1, gang, tang, 0.0)
    scene = syn.make_scene(nlayer, nwno, seed=3)
    scene["F0PI"] = np.ones(nwno)
    scene["surf_reflect"] = np.zeros(nwno)
    keys = resident.REFLECTED_PLANES + ("F0PI", "surf_reflect")
    d = resident.upload_scene(scene, keys, ctx=ctx)
    alb = device.DeviceArray((nwno,), ctx)
    xa = device.DeviceArray((ng, 1, nwno), ctx)
    xb = device.DeviceArray((ng + 1, 1, nwno), ctx)

    def part(c, lo, hi, xint):
        n = hi - lo
        pl = {k: d[k].addr + 8 * lo for k in resident.REFLECTED_PLANES}
        resident.reflected_1d(c, nlevel, n, ng, 1, pl, d["surf_reflect"].addr + 8 * lo, ubar0, ubar1, 1.0,
                              d["F0PI"].addr + 8 * lo, 3, 0, *TTHG, xint.addr, toon_coefficients=0, b_top=0.0,
                              gweight=gw, tweight=tw, albedo=alb.addr + 8 * lo, plane_pitch=nwno)

    def step(n1, g):
        if n1 >= nwno:
            os.environ["PICASO_AMD_ANGLE_GROUP"] = "0"
            part(ctx, 0, nwno, xa)
            return
        _lib.ctx_wait(aux, ctx)
        os.environ["PICASO_AMD_ANGLE_GROUP"] = "0"
        part(ctx, 0, n1, xa)
        os.environ["PICASO_AMD_ANGLE_GROUP"] = str(g)
        part(aux, n1, nwno, xb)
        _lib.ctx_wait(ctx, aux)

    def run(n1, g, n):
        device.timer_start(ctx)
        for _ in range(n):
            step(n1, g)
        return device.timer_stop(ctx) / n

    # reference result
    step(nwno, 0)
    device.sync(ctx)
    ref = alb.to_host()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        run(nwno, 0, 20)
    out = []
    for g in [int(x) for x in args.groups.split(",")]:
        for nb in [int(x) for x in args.blocks.split(",")]:
            n1 = min(nwno, nb * 256)
            run(n1, g, 100)
            ms = [run(n1, g, args.steps) for _ in range(3)]
            device.sync(ctx)
            device.sync(aux)
            same = bool(np.array_equal(alb.to_host(), ref))
            rec = {"g": g, "fused_blocks": nb, "ms": [round(m, 4) for m in ms], "bit_identical": same}
            print(json.dumps(rec), flush=True)
            out.append(rec)
            run(nwno, 0, 50)


if __name__ == "__main__":
    main()
