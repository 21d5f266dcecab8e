#!/usr/bin/env python
"""Experiment (round 6): a balanced-fill shape of the 1e5-column headline launch, measured with the
kernels that exist.  The single launch is 1 563 five-angle waves for 1 024 SIMDs x 2 slots: 540 SIMDs carry two waves,
484 one.  Here the first N1 columns run the fused five-angle kernel (N1 = 65 536: one wave per SIMD) on one stream and
the remaining columns run TWICE at the same time on two more streams -- once with the first three disk angles in the
lane (k_reflected_toa<3>), once with the last two (<2>) -- so that every SIMD would hold a five-angle wave and one
short wave: the cost model says (C + 5A) + (C + 3A) against 2 (C + 5A).  The per-angle arithmetic is grouping-
invariant, so the split columns' intensities are bit-identical to the fused launch's (checked).  The disk sum of the
split columns is left out (xint only): the measured time is a LOWER bound of what a single launch with per-workgroup
roles could reach.  `control`: the same three-stream scaffolding (two event waits per step) around the plain fused
launch and two 64-column launches -- what the scaffolding itself costs.

    python tools/balanced_fill.py [--steps 300]        (GPU box; one JSON line per shape)
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
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--n1", default="65536,57344,61440,69632,73728,81920")
    args = ap.parse_args()
    os.environ["PICASO_AMD_ANGLE_GROUP"] = "0"          # every call: all of its angles in one lane
    ctx, s3, s2 = _lib.context(0), _lib.new_context(0), _lib.new_context(0)
    ng, nwno, nlayer = 5, args.nwno, 90
    nlevel = nlayer + 1
    gang, gw, tang, tw = disco.get_angles_1d(ng)
    ubar0, ubar1, _, _, _ = disco.compute_disco(ng, 1, gang, tang, 0.0)
    scene = syn.make_scene(nlayer, nwno, seed=3)
    scene["F0PI"] = np.ones(nwno)
    scene["surf_reflect"] = np.zeros(nwno)
    d = resident.upload_scene(scene, resident.REFLECTED_PLANES + ("F0PI", "surf_reflect"), ctx=ctx)
    alb = device.DeviceArray((nwno,), ctx)
    x5 = device.DeviceArray((5, 1, nwno), ctx)

    def part(c, lo, hi, a0, a1, xint, fuse):
        n, na = hi - lo, a1 - a0
        pl = {k: d[k].addr + 8 * lo for k in resident.REFLECTED_PLANES}
        resident.reflected_1d(c, nlevel, n, na, 1, pl, d["surf_reflect"].addr + 8 * lo, ubar0[a0:a1], ubar1[a0:a1], 1.0,
                              d["F0PI"].addr + 8 * lo, 3, 0, *TTHG, xint.addr, toon_coefficients=0, b_top=0.0,
                              gweight=gw[a0:a1] if fuse else None, tweight=tw if fuse else None,
                              albedo=alb.addr + 8 * lo if fuse else None, plane_pitch=nwno)

    def fused():
        part(ctx, 0, nwno, 0, 5, x5, True)

    def split(n1, xa, xb, xc, lo2=None):
        """<5> on [0, n1) | <3> and <2> on [lo2 or n1, nwno), all three at once"""
        lo2 = n1 if lo2 is None else lo2
        _lib.ctx_wait(s3, ctx)
        _lib.ctx_wait(s2, ctx)
        part(ctx, 0, n1, 0, 5, xa, True)
        part(s3, lo2, nwno, 0, 3, xb, False)
        part(s2, lo2, nwno, 3, 5, xc, False)
        _lib.ctx_wait(ctx, s3)
        _lib.ctx_wait(ctx, s2)

    def timeit(fn, n):
        device.timer_start(ctx)
        for _ in range(n):
            fn()
        return device.timer_stop(ctx) / n

    fused()
    device.sync(ctx)
    ref = x5.to_host()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        timeit(fused, 20)
    base = [round(timeit(fused, args.steps), 4) for _ in range(3)]
    print(json.dumps({"shape": "fused <5> x 100000 (the headline launch)", "ms": base}), flush=True)
    # control: the scaffolding around the whole fused launch + two one-wave launches
    xa = device.DeviceArray((5, 1, nwno), ctx)
    xb = device.DeviceArray((3, 1, 64), ctx)
    xc = device.DeviceArray((2, 1, 64), ctx)
    ctl = lambda: split(nwno, xa, xb, xc, lo2=nwno - 64)       # noqa: E731
    timeit(ctl, 50)
    print(json.dumps({"shape": "control: fused <5> x 100000 + <3>, <2> x 64 on two more streams (scaffolding cost)",
                      "ms": [round(timeit(ctl, args.steps), 4) for _ in range(3)]}), flush=True)
    for n1 in [int(v) for v in args.n1.split(",")]:
        n2 = nwno - n1
        xa = device.DeviceArray((5, 1, n1), ctx)
        xb = device.DeviceArray((3, 1, n2), ctx)
        xc = device.DeviceArray((2, 1, n2), ctx)
        fn = lambda: split(n1, xa, xb, xc)                      # noqa: E731
        timeit(fn, 50)
        ms = [round(timeit(fn, args.steps), 4) for _ in range(3)]
        for c in (ctx, s3, s2):
            device.sync(c)
        same = bool(np.array_equal(xa.to_host(), ref[:, :, :n1]) and np.array_equal(xb.to_host(), ref[:3, :, n1:])
                    and np.array_equal(xc.to_host(), ref[3:, :, n1:]))
        waves = (-(-n1 // 64), -(-n2 // 64), -(-n2 // 64))
        print(json.dumps({"shape": "<5> x %d | <3> x %d | <2> x %d" % (n1, n2, n2), "waves_5_3_2": waves,
                          "ms": ms, "intensities_bit_identical_to_fused": same}), flush=True)
        timeit(fused, 50)
    print(json.dumps({"shape": "fused <5> x 100000 again", "ms": [round(timeit(fused, args.steps), 4) for _ in range(3)]}),
          flush=True)


if __name__ == "__main__":
    main()
