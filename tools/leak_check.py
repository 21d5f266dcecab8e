#!/usr/bin/env python
"""Retrieval-shaped soak of the product call (run on the GPU box): thousands of spectrum() calls in which EVERY call has
new inputs -- temperature profile, abundances, cloud tables, stellar spectrum, phase geometry, solver family -- so that
every content-keyed cache of the library (resident vectors, cloud tables, opacity shards, climate vectors) sees a new
key each time.  Host RSS and free device memory are sampled along the way: a cache without a bound, an event or a pool
block that is not returned shows up as a slope.  Prints one JSON line; exit code 1 when either grows after the warm-up.

    python tools/leak_check.py [--calls 3000] [--nwno 20000]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from picaso_amd import _lib  # noqa: E402
from picaso_amd import justdoit as jdi  # noqa: E402
from picaso_amd import optics as px  # noqa: E402


def rss_mb():
    with open("/proc/self/statm") as fh:
        return int(fh.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=3000)
    ap.add_argument("--nwno", type=int, default=20000)
    ap.add_argument("--cycles-only", action="store_true",
                    help="pass / fail on device arrays left to the cycle collector only (short runs: the bounded caches "
                         "are still filling)")
    args = ap.parse_args(argv)
    ctx = _lib.context(0)
    hip = ctypes.CDLL("libamdhip64.so")

    def dev_free_mb():
        f, t = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return f.value / 1e6

    def pool():
        o = (ctypes.c_size_t * 6)()
        _lib.check(_lib.load().picaso_ctx_mem_stats(ctx, o), ctx)
        return [round(o[0] / 1e6, 1), int(o[1]), round(o[2] / 1e6, 1), int(o[3]), round(o[4] / 1e6, 1), int(o[5])]

    nwno, nlevel = args.nwno, 61
    wno = np.linspace(2000.0, 33333.0, nwno)
    temps, press = [100.0, 300.0, 700.0, 1500.0, 3000.0], [1e-6, 1e-4, 1e-2, 1e-1, 1.0, 10.0, 100.0, 500.0]
    pt = [(i + 1, p, t) for i, (t, p) in enumerate((t, p) for t in temps for p in press)]
    mols = ["H2O", "CH4", "CO", "NH3", "H2"]
    molecular = {m: {i: 10.0 ** (-24.0 + 2.0 * np.sin(wno / 2500.0 + k) + 0.4 * np.log10(p) + 0.8 * np.log10(t / 300.0))
                     for (i, p, t) in pt} for k, m in enumerate(mols)}
    cia_t = [75.0, 200.0, 500.0, 1000.0, 2000.0, 4000.0]
    continuum = {pr: {t: 10.0 ** (-7.0 + np.cos(wno / 4000.0 + k) + 0.3 * np.log10(t / 300.0)) for t in cia_t}
                 for k, pr in enumerate(("H2H2", "H2He"))}
    ray = {m: 1e-27 * (wno / 1e4) ** 4 for m in ("H2", "He")}
    opa = px.RetrieveOpacities(wno, pt, molecular, continuum, cia_t, rayleigh_opa=ray, query_method="linear", ctx=ctx)
    plev = np.logspace(-6, 2, nlevel)
    rng = np.random.default_rng(11)
    # premixed correlated-k tables (661 bins x 8 Gauss points), as bench.py's product.correlated_k
    nb, nk = 661, 8
    wck = np.linspace(40.0, 28000.0, nb)
    xg, wg = np.polynomial.legendre.leggauss(4)
    gpts = np.concatenate([0.95 * 0.5 * (xg + 1), 0.95 + 0.05 * 0.5 * (xg + 1)])
    gwts = np.concatenate([0.95 * 0.5 * wg, 0.05 * 0.5 * wg])
    tk, pk = np.array(temps), np.array(press)
    lnk = np.log(10.0) * (-26.0 + 2.0 * np.sin(wck / 2500.0)[None, None, :, None] + 0.5 * np.log10(pk)[:, None, None, None]
                          + 0.9 * np.log10(tk / 300.0)[None, :, None, None] + 0.6 * np.arange(nk)[None, None, None, :])
    cont = {pr: {t: 10.0 ** (-7.0 + np.cos(wck / 4000.0 + k) + 0.3 * np.log10(t / 300.0)) for t in cia_t}
            for k, pr in enumerate(("H2H2", "H2He"))}
    opk = px.RetrieveCKs(wck, gwts, np.tile(pk, tk.size), np.repeat(tk, pk.size), np.full(tk.size, pk.size), lnk,
                         continuum=cont, cia_temps=cia_t, rayleigh_opa={m: 1e-27 * (wck / 1e4) ** 4 for m in ("H2", "He")},
                         gauss_pts=gpts, ctx=ctx)
    NKIND = 14
    kinds = {}

    def one(i):
        """call i: a fresh case with inputs nobody has seen before"""
        prof = {"pressure": plev,
                "temperature": (150.0 + 1200.0 * ((np.log10(plev) + 6) / 8) ** 2) * (1.0 + 0.2 * rng.random()),
                "H2": np.full(nlevel, 0.84), "He": np.full(nlevel, 0.155),
                "H2O": np.full(nlevel, 1e-3 * (1 + rng.random())), "CH4": np.full(nlevel, 5e-4 * (1 + rng.random())),
                "CO": np.full(nlevel, 1e-4), "NH3": np.full(nlevel, 1e-5)}
        case = jdi.inputs()
        kind = i % NKIND
        if kind == 6:                # spectrum_batch: three atmospheres of a retrieval's batch
            cases = []
            for _ in range(3):
                c = jdi.inputs()
                c.phase_angle(0)
                c.gravity(gravity=2500.0)
                c.atmosphere(df=dict(prof, temperature=prof["temperature"] * (1.0 + 0.05 * rng.random())))
                c.approx(raman="none")
                cases.append(c)
            rb = jdi.spectrum_batch(cases, opa, calculation="reflected+thermal", batch_size=3)
            kinds[kind] = kinds.get(kind, 0) + 1
            return all(np.all(np.isfinite(r["albedo"])) and np.all(np.isfinite(r["thermal"])) for r in rb)
        if kind == 13:               # spectrum_async: three spectra in flight (the third is never read: its slot is finished on reuse)
            hs = []
            for _ in range(3):
                c = jdi.inputs()
                c.phase_angle(0)
                c.gravity(gravity=2500.0)
                c.atmosphere(df=dict(prof, temperature=prof["temperature"] * (1.0 + 0.05 * rng.random())))
                c.approx(raman="none")
                hs.append(c.spectrum_async(opa, calculation="reflected+thermal"))
            ra = [h.result() for h in hs[:2]]
            kinds[kind] = kinds.get(kind, 0) + 1
            return all(np.all(np.isfinite(r["albedo"])) and np.all(np.isfinite(r["thermal"])) for r in ra)
        if kind in (9, 10):          # k-tables: Toon and SH4 inside the Gauss loop; 11: an 8-phase... a 2-phase thermal curve
            c = jdi.inputs()
            c.phase_angle(0)
            c.gravity(gravity=2500.0)
            c.atmosphere(df=prof)
            c.approx(**({"raman": "none"} if kind == 9 else {"raman": "none", "rt_method": "SH", "stream": 4}))
            r = c.spectrum(opk, calculation="reflected+thermal")
            kinds[kind] = kinds.get(kind, 0) + 1
            return bool(np.all(np.isfinite(r["albedo"])) and np.all(np.isfinite(r["thermal"])))
        if kind == 11:
            pert = 1.0 + 0.1 * np.cos(np.arange(16).reshape(4, 4) + rng.random())
            pc = jdi.inputs()
            pc.phase_curve_geometry("thermal", [0.4, 2.0], num_gangle=4, num_tangle=4)
            pc.gravity(gravity=2500.0)
            pc.atmosphere_4d([dict(prof, temperature=prof["temperature"][:, None, None] * (pert[None] + 0.01 * k))
                              for k in range(2)])
            pc.approx(raman="none")
            curve = pc.phase_curve(opk)
            kinds[kind] = kinds.get(kind, 0) + 1
            return all(np.all(np.isfinite(v["thermal"])) for v in curve.values())
        if kind == 7:                # 3-D: 4 x 4 facets, per-facet temperatures, a cloud map on its own grid every other call
            pert = 1.0 + 0.1 * np.cos(np.arange(16).reshape(4, 4) + rng.random())
            c3 = jdi.inputs()
            c3.phase_angle(np.pi / 3, num_gangle=4, num_tangle=4)
            c3.gravity(gravity=2500.0)
            c3.atmosphere_3d(dict(prof, temperature=prof["temperature"][:, None, None] * pert[None]))
            c3.approx(raman="none")
            if (i // 9) % 2:
                box = np.zeros((nlevel - 1, 196))
                box[30:40] = rng.uniform(0.1, 0.5)
                c3.clouds_3d(df={"opd": box, "w0": np.where(box > 0, 0.95, 0.0), "g0": np.where(box > 0, 0.6, 0.0),
                                 "wavenumber": np.linspace(wno[0], wno[-1], 196)})
            r = c3.spectrum(opa, calculation="reflected+thermal", dimension="3d")
            kinds[kind] = kinds.get(kind, 0) + 1
            return bool(np.all(np.isfinite(r["albedo"])) and np.all(np.isfinite(r["thermal"])))
        if kind == 4:
            case.phase_angle(0.3 + rng.random(), num_gangle=6, num_tangle=6)
        else:
            case.phase_angle(0)
        case.gravity(gravity=2500.0 * (1 + 0.1 * rng.random()))
        case.atmosphere(df=prof)
        akw = {"raman": "none"}
        if kind == 3:
            akw.update(rt_method="SH", stream=4)
        if kind == 5:
            akw.update(get_lvl_flux=True)
        case.approx(**akw)
        calc = "reflected+thermal"
        hk = dict(do_holes=True, fhole=0.3, fthin_cld=0.1) if kind == 8 else {}     # patchy cloud: two plane sets, blended
        if kind in (1, 2, 3, 8):     # cloud tables on the opacity grid, new numbers every call
            shp = (nlevel - 1, nwno)
            opd = np.zeros(shp)
            top = int(rng.integers(20, 45))
            opd[top:top + 6] = rng.uniform(0.05, 1.0)
            case.clouds(df={"opd": opd, "w0": np.full(shp, rng.uniform(0.5, 0.99)), "g0": np.full(shp, rng.uniform(0.1, 0.8))}, **hk)
        if kind == 2:
            case.star(relative_flux=1.0 + 0.2 * np.cos(wno / (800.0 + 200.0 * rng.random())), radius=6.9e10,
                      semi_major=7.5e12)
            case.gravity(radius=7.1e9, mass=1.9e30)
            calc = "reflected+thermal+transmission"
        r = case.spectrum(opa, calculation=calc, devices=[0, 0] if kind == 12 else None)   # 12: two wavelength blocks
        kinds[kind] = kinds.get(kind, 0) + 1
        ok = all(np.all(np.isfinite(v)) for v in r.values() if isinstance(v, np.ndarray) and v.dtype == np.float64)
        return ok

    samples = []
    t0 = time.perf_counter()
    finite = True
    every = max(1, args.calls // 30)
    assert args.calls >= NKIND
    for i in range(args.calls):
        finite = one(i) and finite
        if i % every == 0 or i == args.calls - 1:
            samples.append((i, round(rss_mb(), 1), round(dev_free_mb(), 1), pool()))
    dt = time.perf_counter() - t0
    # what is alive at the end, by shape, before and after a collection (who keeps it: the first non-trivial referrer)
    import collections
    import gc
    from picaso_amd import device

    def census():
        """every live DeviceArray by shape -- the FULL counter: a top-N cut makes a shape that is tied at rank N appear to
        vanish when gc.get_objects() changes its order (round 5's red GPUTEST: five resident (nwno,) vectors, nothing cyclic)"""
        c = collections.Counter()
        for o in gc.get_objects():
            if isinstance(o, device.DeviceArray):
                c[str(tuple(o.shape))] += 1
        return dict(c)

    def holders():
        """Who keeps the arrays only a collection frees: collect with DEBUG_SAVEALL (nothing is freed, everything
        unreachable lands in gc.garbage) and name, for each unreachable DeviceArray, its referrers two levels up."""
        def name(o):
            t = type(o).__name__
            if isinstance(o, dict):
                return "dict(keys=%s)" % sorted(map(str, o))[:8]
            if t in ("function", "method"):
                return "%s %s" % (t, getattr(o, "__qualname__", "?"))
            if t == "frame":
                return "frame %s:%d" % (o.f_code.co_qualname if hasattr(o.f_code, "co_qualname") else o.f_code.co_name, o.f_lineno)
            if t == "cell":
                return "cell"
            if isinstance(o, (list, tuple)):
                return "%s[%d]" % (t, len(o))
            return t
        gc.set_debug(gc.DEBUG_SAVEALL)
        try:
            gc.collect()
            junk = list(gc.garbage)
        finally:
            gc.set_debug(0)
            del gc.garbage[:]
        ids = {id(o) for o in junk}
        chains = collections.Counter()
        for o in junk:
            if not isinstance(o, device.DeviceArray):
                continue
            for r1 in gc.get_referrers(o):
                if id(r1) not in ids:
                    continue
                ups = sorted({name(r2) for r2 in gc.get_referrers(r1) if id(r2) in ids})[:4]
                chains["%s <- %s <- %s" % (tuple(o.shape), name(r1), " | ".join(ups))] += 1
        del junk
        return dict(chains.most_common(40))
    alive_before = census()
    held_by = holders()          # (frees nothing: SAVEALL keeps the unreachable objects alive until the list is cleared)
    gc.collect()
    alive_after = census()
    # after the first third (pools, rings, code objects have reached their size) nothing may grow
    third = [s for s in samples if s[0] >= args.calls // 3]
    rss_growth = third[-1][1] - third[0][1]
    dev_growth = third[0][2] - third[-1][2]
    out = {"calls": args.calls, "nwno": nwno, "seconds": round(dt, 1), "ms_per_call": round(1e3 * dt / args.calls, 3),
           "calls_by_kind": kinds, "device_arrays_alive_at_end": alive_before, "after_gc_collect": alive_after, "finite": bool(finite),
           "host_rss_mb_growth_after_first_third": round(rss_growth, 1),
           "device_mb_growth_after_first_third": round(dev_growth, 1), "samples_call_rss_devfree_[liveMB,liveN,poolMB,poolN,pinnedMB,pinnedN]": samples}
    cyclic = {k: v - alive_after.get(k, 0) for k, v in alive_before.items() if v != alive_after.get(k, 0)}
    out["device_arrays_that_waited_for_the_cycle_collector"] = cyclic
    out["held_by"] = held_by
    if cyclic:                   # the one thing a reader of a failure needs comes last (the caller prints the tail)
        print(json.dumps({"cyclic": cyclic, "held_by": held_by}), file=sys.stderr)
    print(json.dumps(out))
    return 0 if (finite and not cyclic and (args.cycles_only or dev_growth < 256.0)) else 1


if __name__ == "__main__":
    sys.exit(main())
