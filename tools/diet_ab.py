#!/usr/bin/env python
"""A/B of the round-5 regrouping of the reflected layer body (PZ_REFL_DIET, common.hpp) on the headline scene: the library as
built against a second build with -DPZ_REFL_DIET=0 (the round-4 operation order: 1-ulp reciprocals, no shared products).

    python tools/diet_ab.py --build            # here (no GPU needed): hipcc the second library into picaso_amd/build_ab/
    python tools/diet_ab.py [-o out.json]      # GPU box: both libraries on BASELINE configs[2]'s scene, in child processes

Prints the largest relative difference of the 1e5 albedos and of the 5 x 1e5 intensities, split into the 2 000 columns closest
to the two-stream singularity lambda^2 = 1/ubar0^2 -- where the reference's own fp64 result carries 1e-9 of rounding
(tools/headline_error_x87.py) and ANY regrouping of the arithmetic moves the last digits by a like amount -- and the other
98 000.  Round 5's advisor guessed <= 2e-13 for the whole scene; measured (profiles/r06_diet_ab.json): that holds away from the
singularity, and the ill-conditioned columns move by up to 2e-10 in the albedo, inside the 1e-9 the tests hold against the
oracle there and 4 000 x inside BASELINE's 1e-6.  Exit code 1 above 1e-9 anywhere or 2e-12 on the well-conditioned columns."""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
AB_DIR = os.path.join(ROOT, "picaso_amd", "build_ab")
AB_LIB = os.path.join(AB_DIR, "libpicaso_hip_nodiet.so")


def build():
    from picaso_amd import build as b
    os.makedirs(AB_DIR, exist_ok=True)
    objs, procs = [], []
    for src in b.sources():
        obj = os.path.join(AB_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        procs.append((src, subprocess.Popen([b.HIPCC] + b.FLAGS + ["-DPZ_REFL_DIET=0", "-c", src, "-o", obj],
                                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            sys.stderr.write(out.decode())
            raise SystemExit("hipcc failed on %s" % src)
    rocm_lib = os.path.join(b.ROCM, "lib")
    subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", AB_LIB] + objs +
                          ["-L" + rocm_lib, "-lrccl", "-Wl,-rpath," + rocm_lib])
    for o in objs:
        os.remove(o)
    print(AB_LIB)


def solve(out):
    """child: the headline scene through whatever library PICASO_AMD_LIB names"""
    from picaso_amd import _lib, device, disco, resident
    from picaso_amd import synthetic as syn
    nwno, nlayer, ng = 100000, 90, 5
    ctx = _lib.context(0)
    sc = syn.make_scene(nlayer, nwno, seed=3)
    gang, gw, tang, tw = disco.get_angles_1d(ng)
    u0, u1, _, _, _ = disco.compute_disco(ng, 1, gang, tang, 0.0)
    sc["F0PI"], sc["surf_reflect"] = np.ones(nwno), np.zeros(nwno)
    d = resident.upload_scene(sc, resident.REFLECTED_PLANES + ("F0PI", "surf_reflect"), ctx=ctx)
    x, alb = device.DeviceArray((ng, 1, nwno), ctx), device.DeviceArray((nwno,), ctx)
    res = {}
    for tag, n in (("full", nwno), ("shard", 12500)):          # the fused launch and the cooperative kernel of a shard
        resident.reflected_1d(ctx, nlayer + 1, n, ng, 1, d, d["surf_reflect"], u0, u1, 1.0, d["F0PI"], 3, 0, 1.0, -1.0, 2.0,
                              -0.5, 1.0, x, gweight=gw, tweight=tw, albedo=alb, plane_pitch=nwno)
        device.sync(ctx)
        res["alb_" + tag] = alb.to_host()[:n].copy()
        res["x_" + tag] = x.to_host().reshape(-1)[:ng * n].reshape(ng, n).copy()
    # distance to the singularity of the direct-beam particular solution, per column: min over layers and angles
    w0, cb, fc = sc["w0"], sc["cosb"], sc["ftau_cld"]
    g1 = (np.sqrt(3.0) / 2) * (2 - w0 * (1 + fc * cb))
    g2 = (np.sqrt(3.0) * w0 / 2) * (1 - fc * cb)
    lam2 = g1 * g1 - g2 * g2
    res["sing"] = np.min(np.abs(lam2[None] - 1.0 / (u0.reshape(-1)[:, None, None] ** 2)), axis=(0, 1))
    np.savez(out, **res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--child")
    ap.add_argument("-o")
    a = ap.parse_args()
    if a.build:
        return build()
    if a.child:
        return solve(a.child)
    if not os.path.exists(AB_LIB):
        raise SystemExit("no %s: run `python tools/diet_ab.py --build` first (hipcc, no GPU needed)" % AB_LIB)
    outs = {}
    for tag, lib in (("diet", None), ("nodiet", AB_LIB)):
        env = dict(os.environ)
        if lib:
            env["PICASO_AMD_LIB"] = lib
        path = "/tmp/diet_ab_%s.npz" % tag
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", path], env=env)
        outs[tag] = np.load(path)
    A, B = outs["diet"], outs["nodiet"]

    def rel(k, sel=slice(None)):
        a_, b_ = A[k][..., sel], B[k][..., sel]
        return float(np.max(np.abs(a_ - b_) / np.abs(b_)))
    order = np.argsort(A["sing"])
    worst, rest = order[:2000], order[2000:]
    rec = {"scene": "bench.py headline (BASELINE configs[2], seed 3): 1e5 x 90 x 5",
           "max_rel_albedo_full_launch": rel("alb_full"), "max_rel_intensity_full_launch": rel("x_full"),
           "max_rel_albedo_12500_shard_coop": rel("alb_shard"), "max_rel_intensity_12500_shard_coop": rel("x_shard"),
           "max_rel_albedo_2000_columns_nearest_singularity": rel("alb_full", worst),
           "max_rel_albedo_other_98000_columns": rel("alb_full", rest),
           "max_rel_intensity_other_98000_columns": rel("x_full", rest),
           "median_rel_albedo": float(np.median(np.abs(A["alb_full"] - B["alb_full"]) / np.abs(B["alb_full"]))),
           "singularity_distance_of_column_2000": float(A["sing"][order[2000]]),
           "min_singularity_distance": float(A["sing"][worst[0]]),
           "diet_full_equals_diet_shard_bits": bool(np.array_equal(A["alb_full"][:12500], A["alb_shard"])),
           "nodiet_full_equals_nodiet_shard_bits": bool(np.array_equal(B["alb_full"][:12500], B["alb_shard"])),
           "bounds": {"anywhere_albedo": 1e-9, "well_conditioned_albedo": 2e-12}}
    rec["within_bounds"] = bool(rec["max_rel_albedo_full_launch"] <= 1e-9 and rec["max_rel_albedo_12500_shard_coop"] <= 1e-9
                                and rec["max_rel_albedo_other_98000_columns"] <= 2e-12)
    print(json.dumps(rec))
    if a.o:
        with open(a.o, "w") as fh:
            json.dump(rec, fh, indent=1)
    return 0 if rec["within_bounds"] else 1


if __name__ == "__main__":
    sys.exit(main())
