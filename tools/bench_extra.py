#!/usr/bin/env python
"""Secondary measurements quoted in DESIGN.md (run on the GPU box): thermal (BASELINE configs[1]),
3-D reflected facets, level-flux kernels, opacity pre-stage, and the PCIe-inclusive host-pointer
call of the headline workload.  HIP-event timed on the library's stream, inputs resident in HBM
unless stated."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from picaso_amd import _lib, device, disco, fluxes, resident  # noqa: E402
from picaso_amd import synthetic as syn  # noqa: E402
from picaso_amd.device import DeviceArray  # noqa: E402

TTHG = (1.0, -1.0, 2.0, -0.5, 1.0)


def timeit(fn, ctx, reps=10, warm=2, prewarm_s=0.25):
    """Mean HIP-event time per call in the GPU's steady clock state: an idle MI355X needs a few hundred
    launches (tens of ms) to ramp its clocks up and drops back within ms of idling, so the timed calls
    follow `prewarm_s` of the same calls without a gap."""
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < prewarm_s or n < warm:
        fn()
        n += 1
        if n % 16 == 0:
            device.sync(ctx)
    device.timer_start(ctx)
    for _ in range(reps):
        fn()
    return device.timer_stop(ctx) / reps


def main():
    ctx = _lib.context(0)
    out = {}
    g, gw, t, tw = disco.get_angles_1d(5)
    u0, u1, ct, _, _ = disco.compute_disco(5, 1, g, t, 0.0)
    nlayer = 90
    only = os.environ.get("BENCH_ONLY")        # thermal | variants | pipeline | 3d | sh | copy | mix | climate | e2e (default: all)
    if only in (None, "thermal"):
        for nwno in (10000, 100000):
            sc = syn.make_scene(nlayer, nwno, seed=5)
            d = resident.upload_scene(sc, ("dtau_og", "w0_no_raman", "cosb_og", "wno"), ctx=ctx)
            rs = DeviceArray.from_host(np.zeros(nwno), ctx)
            flux = DeviceArray((5, 1, nwno), ctx)
            disk = DeviceArray((nwno,), ctx)
            ms = timeit(lambda: resident.thermal_1d(ctx, nlayer + 1, d["wno"], nwno, 5, 1, sc["tlevel"],
                                                    d["dtau_og"], d["w0_no_raman"], d["cosb_og"],
                                                    sc["plevel"], u1, rs, 0, flux, gweight=gw, tweight=tw,
                                                    flux_disk=disk), ctx)
            ab = 8 * nwno * (3 * nlayer + 3 + 5 + 1)
            out["thermal_%d" % nwno] = dict(ms=ms, spectra_per_s=1e3 / ms, GBps=ab / ms / 1e6,
                                            algorithmic_bytes=ab)
            if nwno == 100000:
                # PCIe-inclusive: host-pointer drop-in call of the headline reflected workload
                planes = [sc[k] for k in resident.REFLECTED_PLANES]
                t0 = time.perf_counter()
                fluxes.get_reflected_1d(nlayer + 1, sc["wno"], nwno, 5, 1, *planes, 0.0, u0, u1, 1.0,
                                        np.ones(nwno), 3, 0, *TTHG)
                t1 = time.perf_counter()
                fluxes.get_reflected_1d(nlayer + 1, sc["wno"], nwno, 5, 1, *planes, 0.0, u0, u1, 1.0,
                                        np.ones(nwno), 3, 0, *TTHG)
                t2 = time.perf_counter()
                out["reflected_host_pointers_1e5"] = dict(first_call_s=t1 - t0, second_call_s=t2 - t1,
                                                          spectra_per_s=1.0 / (t2 - t1),
                                                          note="includes np.zeros of the 4 level-flux arrays "
                                                               "(1.46 GB) the reference signature returns")
                # level fluxes (climate caller shape: one angle, ubar = 0.5): the two-sweep kernels
                dd = resident.upload_scene(sc, resident.REFLECTED_PLANES + ("w0_no_raman",), ctx=ctx)
                f0d = DeviceArray.from_host(np.ones(nwno), ctx)
                dwd = DeviceArray.from_host(np.full(nwno, 0.3), ctx)
                xl = DeviceArray((1, 1, nwno), ctx)
                lv4 = [DeviceArray((1, 1, nlayer + 1, nwno), ctx) for _ in range(4)]
                half = np.array([[0.5]])
                ms = timeit(lambda: resident.reflected_1d_ck(
                    ctx, nlayer + 1, nwno, 1, 1, 1, dd, rs, half, half, 1.0, f0d, 3, 0, *TTHG, np.ones(1), xl,
                    get_toa_intensity=0, lvl_fluxes=lv4), ctx)
                lb = 8 * nwno * (9 * nlayer + 2 * (nlayer + 1) + 2 + 4 * (nlayer + 1) + 8 * nlayer)
                out["reflected_lvl_1e5"] = dict(ms=ms, GBps=lb / ms / 1e6,
                                                note="bytes: planes + 4 level outputs + 4-plane sweep scratch written and read")
                ms = timeit(lambda: resident.thermal_1d_ck(
                    ctx, nlayer + 1, d["wno"], nwno, 1, 1, 1, sc["tlevel"], dd["dtau_og"], dd["w0_no_raman"],
                    dd["cosb_og"], sc["plevel"], half, rs, 0, np.ones(1), xl, dwno=dwd, calc_type=1,
                    lvl_fluxes=lv4), ctx)
                lb = 8 * nwno * (3 * nlayer + 3 + 4 * (nlayer + 1) + 8 * nlayer)
                out["thermal_lvl_1e5"] = dict(ms=ms, GBps=lb / ms / 1e6)
    if only in (None, "variants"):
        # Headline workload variants: (a) as bench.py (cloud slab in 10 of 90 layers: the other layers
        # are not delta-scaled and skip the second exponential), (b) cloud in every layer (every layer
        # delta-scaled: no shortcut), (c) phase angle 60 deg (ubar0 != ubar1: general kernel)
        nwno = 100000
        base = syn.make_scene(nlayer, nwno, seed=3)
        comps = [base[k].copy() for k in ("taugas", "tauray", "taucld", "w0_cld", "g0_cld")]
        rng = np.random.default_rng(11)
        comps[2] = comps[2] + 0.01 * comps[0].mean() * (1.0 + 0.2 * rng.random((nlayer, 1)))
        comps[3] = np.where(comps[3] > 0, comps[3], 0.9)
        comps[4] = np.where(comps[4] > 0, comps[4], 0.6)
        everywhere = syn.mix_planes(*comps)
        u0p, u1p, ctp, _, _ = disco.compute_disco(5, 1, g, t, np.pi / 3)
        for tag, planes, a0, a1, cth in (("slab", base, u0, u1, 1.0), ("cloud_everywhere", everywhere, u0, u1, 1.0),
                                         ("phase60", base, u0p, u1p, float(ctp))):
            pl = dict(planes)
            pl["F0PI"] = np.ones(nwno)
            pl["surf_reflect"] = np.zeros(nwno)
            dd = resident.upload_scene(pl, resident.REFLECTED_PLANES + ("F0PI", "surf_reflect"), ctx=ctx)
            xi = DeviceArray((5, 1, nwno), ctx)
            al = DeviceArray((nwno,), ctx)
            ms = timeit(lambda: resident.reflected_1d(ctx, nlayer + 1, nwno, 5, 1, dd, dd["surf_reflect"], a0, a1,
                                                      cth, dd["F0PI"], 3, 0, *TTHG, xi, toon_coefficients=0,
                                                      b_top=0.0, gweight=gw, tweight=tw, albedo=al), ctx, reps=20)
            out["reflected_1e5_%s" % tag] = dict(ms=ms, spectra_per_s=1e3 / ms, frac_of_8TBs=0.8 / ms / 8.0)
    if only in (None, "pipeline"):
        # Independent spectra kept in flight on two streams (two contexts): a 1e5-column spectrum is
        # 1.5 waves per SIMD, so the tail of one launch (half the SIMDs idle) overlaps the head of
        # the next.  Throughput over 40 spectra, same workload as bench.py.
        nwno = 100000
        sc = syn.make_scene(nlayer, nwno, seed=3)
        sc["F0PI"] = np.ones(nwno)
        sc["surf_reflect"] = np.zeros(nwno)
        ctxs = [ctx, _lib.new_context(0), _lib.new_context(0), _lib.new_context(0)]
        dd = resident.upload_scene(sc, resident.REFLECTED_PLANES + ("F0PI", "surf_reflect"), ctx=ctx)
        outs = [(DeviceArray((5, 1, nwno), c), DeviceArray((nwno,), c)) for c in ctxs]
        device.sync(ctx)

        def spectrum(j):
            c = ctxs[j % len(ctxs)]
            xi, al = outs[j % len(ctxs)]
            resident.reflected_1d(c, nlayer + 1, nwno, 5, 1, dd, dd["surf_reflect"], u0, u1, 1.0, dd["F0PI"], 3, 0,
                                  *TTHG, xi, toon_coefficients=0, b_top=0.0, gweight=gw, tweight=tw, albedo=al)
        for nstream in (1, 2, 3, 4):
            use = ctxs[:nstream]
            ctxs_run = use
            for j in range(4):
                c = use[j % nstream]
                resident.reflected_1d(c, nlayer + 1, nwno, 5, 1, dd, dd["surf_reflect"], u0, u1, 1.0, dd["F0PI"],
                                      3, 0, *TTHG, outs[j % nstream][0], gweight=gw, tweight=tw,
                                      albedo=outs[j % nstream][1])
            for c in use:
                device.sync(c)
            t0 = time.perf_counter()
            K = 600
            for j in range(K):
                c = use[j % nstream]
                resident.reflected_1d(c, nlayer + 1, nwno, 5, 1, dd, dd["surf_reflect"], u0, u1, 1.0, dd["F0PI"],
                                      3, 0, *TTHG, outs[j % nstream][0], gweight=gw, tweight=tw,
                                      albedo=outs[j % nstream][1])
            for c in use:
                device.sync(c)
            dt_ = time.perf_counter() - t0
            out["reflected_1e5_throughput_%dstream" % nstream] = dict(ms_per_spectrum=1e3 * dt_ / K,
                                                                      spectra_per_s=K / dt_)
        assert all(np.array_equal(outs[0][1].to_host(), o[1].to_host()) for o in outs[1:])
    import ctypes
    from picaso_amd._lib import check, f64, load, ptr
    ci, cd = ctypes.c_int, ctypes.c_double
    if only in (None, "3d"):
        # 3-D facets: 8x8 facets, 90 layers, 4096 wavelengths (same per-facet planes replicated)
        ng = nt = 8
        nw3 = 4096
        sc = syn.make_scene(nlayer, nw3, seed=7)
        gg, ggw, tt, ttw = disco.get_angles_3d(ng, nt)
        v0, v1, cth, _, _ = disco.compute_disco(ng, nt, gg, tt, np.pi / 3)
        dev3 = {}
        for k in resident.REFLECTED_PLANES:
            dev3[k] = DeviceArray.from_host(np.repeat(sc[k][:, :, None], ng * nt, axis=2), ctx)
        f0 = DeviceArray.from_host(np.ones(nw3), ctx)
        rs3 = DeviceArray.from_host(np.zeros(nw3), ctx)
        x3 = DeviceArray((ng, nt, nw3), ctx)
        a3 = DeviceArray((nw3,), ctx)

        def run3d():
            check(load().picaso_get_reflected_3d_dev(
                ctx, ci(nlayer + 1), ci(nw3), ci(ng), ci(nt), *[ptr(dev3[k].addr) for k in resident.REFLECTED_PLANES],
                ptr(rs3.addr), ptr(f64(v0)), ptr(f64(v1)), cd(cth), ptr(f0.addr), ci(0), ci(0),
                *[cd(v) for v in TTHG], ptr(x3.addr), ptr(f64(ggw)), ptr(f64(ttw)), ptr(a3.addr)), ctx)
        ms = timeit(run3d, ctx, reps=5)
        ab = 8 * nw3 * ng * nt * (9 * nlayer + 2 * (nlayer + 1) + 1)
        out["reflected_3d_8x8_%d" % nw3] = dict(ms=ms, GBps=ab / ms / 1e6, algorithmic_bytes=ab,
                                                facet_columns_per_s=nw3 * ng * nt / ms * 1e3)
    if only in (None, "sh"):
        # SH4 reflected (BASELINE configs[3] per-GPU shard sizes): 12 500 and 100 000 wavelengths
        for nwno in (12500, 100000):
            sc = syn.make_scene(nlayer, nwno, seed=9, stream=4)
            names = ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray", "f_deltaM", "dtau_og", "tau_og",
                     "w0_og", "cosb_og")
            dd = resident.upload_scene(sc, names, ctx=ctx)
            f0 = DeviceArray.from_host(np.ones(nwno), ctx)
            rs = DeviceArray.from_host(np.zeros(nwno), ctx)
            x = DeviceArray((5, 1, nwno), ctx)
            alb = DeviceArray((nwno,), ctx)

            def runsh():
                check(load().picaso_get_reflected_SH_dev(
                    ctx, ci(nlayer + 1), ci(nwno), ctypes.c_long(nwno), ci(5), ci(1),
                    *[ptr(dd[k].addr) for k in names], ptr(rs.addr), ptr(f64(u0)), ptr(f64(u1)), cd(1.0),
                    ptr(f0.addr), ci(0), ci(0), ci(0), ci(1), ci(1), ci(1), *[cd(v) for v in TTHG], ci(4),
                    cd(0.0), ci(0), ci(0), ci(1), ptr(x.addr), None, ptr(f64(gw)), ptr(f64(tw)), ptr(alb.addr)), ctx)
            ms = timeit(runsh, ctx, reps=5)
            ab = 8 * nwno * (9 * nlayer + 2 * (nlayer + 1) + 2 + 5 + 1)
            out["reflected_SH4_%d" % nwno] = dict(ms=ms, spectra_per_s=1e3 / ms, GBps_algorithmic=ab / ms / 1e6)
    if only in (None, "copy"):
        # measured HBM ceiling on this box: device-to-device copy of 2 GiB (read + write = 4 GiB of traffic)
        import ctypes as _ct
        from picaso_amd._lib import check as _check, load as _load
        nbytes = 2 << 30
        src, dst = DeviceArray((nbytes // 8,), ctx), DeviceArray((nbytes // 8,), ctx)
        src.zero()
        ms = timeit(lambda: _check(_load().picaso_memcpy_d2d(ctx, _ct.c_void_p(dst.addr), _ct.c_void_p(src.addr),
                                                              _ct.c_size_t(nbytes)), ctx), ctx, reps=10)
        out["hbm_copy_d2d_2GiB"] = dict(ms=ms, GBps_read_plus_write=2 * nbytes / ms / 1e6)
        src.free(); dst.free()
    if only in (None, "mix"):
        # on-the-fly correlated-k mixing at the climate tables' shape: 661 bins x 8 Gauss points,
        # 90 layers x 4 P-T neighbours, 17 gases; CPU oracle on a 3-layer sample of the same job
        from oracle import oracle as orc
        rng = np.random.default_rng(11)
        nk, ngas, npres, ntemp, nw, nl = 8, 17, 20, 15, 661, 90
        xg, wg = np.polynomial.legendre.leggauss(4)
        pts = np.concatenate([0.95 * 0.5 * (xg + 1), 0.95 + 0.05 * 0.5 * (xg + 1)])
        wts = np.concatenate([0.95 * 0.5 * wg, 0.05 * 0.5 * wg])
        kap = [-55.0 + 10.0 * rng.random((npres, ntemp, nw, 1))
               + np.cumsum(rng.random((npres, ntemp, nw, nk)) * rng.choice([0.1, 1.0, 5.0]), axis=3)
               for _ in range(ngas)]
        mixes = [10.0 ** (-1.0 - 7.0 * rng.random(nl)) for _ in range(ngas)]
        p_low, t_low = rng.integers(0, npres - 1, nl), rng.integers(0, ntemp - 1, nl)
        idx = np.array([p_low, p_low + 1, t_low, t_low + 1])
        dk = [DeviceArray.from_host(k, ctx) for k in kap]
        outm = DeviceArray((nl, 4, nw, nk), ctx)
        ms = timeit(lambda: resident.mix_all_gases_gasesfly(ctx, dk, mixes, pts, wts, idx, out=outm), ctx, reps=5)
        t0 = time.perf_counter()
        ref = orc.mix_all_gases_gasesfly(kap, [m[:3] for m in mixes], pts, wts, idx[:, :3])
        cpu_s = (time.perf_counter() - t0) * nl / 3
        err = float(np.max(np.abs(np.moveaxis(outm.to_host()[:3], 1, 3) - ref)))
        pair_mixes = nl * 4 * nw * (ngas - 1)
        out["ckmix_661x8_90layers_17gases"] = dict(ms=ms, pair_mixes_per_s=pair_mixes / ms * 1e3,
                                                   cpu_oracle_1core_s=cpu_s, speedup=cpu_s * 1e3 / ms,
                                                   max_abs_err_lnk_vs_oracle=err)
        for d_ in dk:
            d_.free()
    if only in (None, "climate"):
        # climate.get_fluxes at the climate tables' shape: 91 levels, 661 bins x 8 Gauss points, 5 disk
        # angles for the thermal leg; host-array call (PCIe inclusive), planes already resident, CPU oracle
        from oracle import climate_oracle as co
        from picaso_amd import climate as pc
        nlev, nw, ngq = 91, 661, 8
        scs = [syn.make_scene(nlev - 1, nw, seed=70 + ig, gas_scale=10.0 ** (0.5 * ig - 2)) for ig in range(ngq)]
        keys = resident.REFLECTED_PLANES + ("w0_no_raman",)
        st = {k: np.ascontiguousarray(np.stack([sc[k] for sc in scs], axis=2)) for k in keys}
        xg, wg = np.polynomial.legendre.leggauss(ngq)
        wno_c = scs[0]["wno"]
        atm_t = pc.Atmosphere_Tuple(None, None, nlev, scs[0]["tlevel"], scs[0]["plevel"], None, None, None, None)
        sp_t = pc.ScatteringPhase_Tuple(np.zeros(nw), 3, 0, 1.0, -1.0, 2.0, -0.5, 1.0)
        dis_t = pc.Disco_Tuple(5, 1, gw, tw, u0, u1, 1.0)
        og_t = pc.Opagrid_Tuple(nw, np.abs(np.gradient(wno_c)), wno_c, ngq, 0.5 * wg)

        def tuples(conv):
            return (pc.OpacityWEd_Tuple(*[conv(st[k]) for k in ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray",
                                                                 "gcos2", "w0_no_raman")], None),
                    pc.OpacityNoEd_Tuple(*[conv(st[k]) for k in ("dtau_og", "tau_og", "w0_og", "cosb_og")]))
        wh, nh = tuples(lambda a: a)
        wd, nd = tuples(lambda a: DeviceArray.from_host(a, ctx))
        res = {}
        for tag, (w_, n_) in (("host_arrays", (wh, nh)), ("resident_planes", (wd, nd))):
            pc.get_fluxes(atm_t, w_, n_, sp_t, dis_t, og_t, np.ones(nw), True, True)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                pc.get_fluxes(atm_t, w_, n_, sp_t, dis_t, og_t, np.ones(nw), True, True)
                ts.append(time.perf_counter() - t0)
            res[tag + "_ms"] = 1e3 * min(ts)
        t0 = time.perf_counter()
        co.get_fluxes(atm_t, wh, nh, sp_t, dis_t, og_t, np.ones(nw), True, True)
        res["cpu_oracle_1core_ms"] = 1e3 * (time.perf_counter() - t0)
        # the Jacobian of the T(P) iteration (reference climate.py:1105-1180): one thermal get_fluxes per perturbed level
        # -- as a loop of calls and as ONE get_fluxes_tbatch call
        tl0 = np.asarray(scs[0]["tlevel"], dtype=float)
        temps = np.stack([tl0 + (np.arange(nlev) == jm) * max(1e-4 * tl0[jm], 3.0) for jm in range(nlev)])
        pc.get_fluxes_tbatch(temps[:4], atm_t, wd, nd, sp_t, dis_t, og_t, ctx=ctx)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            jb = pc.get_fluxes_tbatch(temps, atm_t, wd, nd, sp_t, dis_t, og_t, ctx=ctx)
            ts.append(time.perf_counter() - t0)
        res["jacobian_%d_profiles_tbatch_ms" % nlev] = 1e3 * min(ts)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            jn = pc.get_fluxes_tbatch(temps, atm_t, wd, nd, sp_t, dis_t, og_t, ctx=ctx, chunk=nlev, nets_only=True)
            ts.append(time.perf_counter() - t0)
        res["jacobian_%d_profiles_tbatch_nets_only_ms" % nlev] = 1e3 * min(ts)
        res["jacobian_nets_only_max_rel_diff"] = float(max(np.max(np.abs(jn[j] - jb[j]) / np.abs(jb[j]).max()) for j in range(2)))
        t0 = time.perf_counter()
        jl = [pc.get_fluxes(atm_t._replace(t_level=t), wd, nd, sp_t, dis_t, og_t, np.ones(nw), False, True, ctx=ctx)[4:] for t in temps]
        res["jacobian_%d_profiles_loop_ms" % nlev] = 1e3 * (time.perf_counter() - t0)
        res["jacobian_tbatch_equals_loop"] = bool(all(np.array_equal(jb[j][k], jl[k][j]) for k in range(nlev) for j in range(4)))
        out["climate_get_fluxes_91x661x8"] = res
    if only in (None, "e2e"):
        # inputs.spectrum() end to end at 1e5 wavelengths x 90 layers: HBM-resident synthetic opacity
        # tables (5 molecules x 40 (P,T) points, 2 CIA pairs), linear interpolation, cloud slab
        from picaso_amd import justdoit as jdi
        from picaso_amd import optics as px
        nwno = 100000
        wno = np.linspace(2000.0, 33333.0, nwno)
        rng = np.random.default_rng(1)
        temps = [100.0, 300.0, 700.0, 1500.0, 3000.0]
        press = [1e-6, 1e-4, 1e-2, 1e-1, 1.0, 10.0, 100.0, 500.0]
        pt, pid = [], 0
        for t_ in temps:
            for p_ in press:
                pid += 1
                pt.append((pid, p_, t_))
        mols = ["H2O", "CH4", "CO", "NH3", "H2"]
        molecular = {m: {i: 10.0 ** (-24.0 + 2.0 * np.sin(wno / 2500.0 + k) + 0.4 * np.log10(p_) + 0.8 * np.log10(t_ / 300.0))
                         for (i, p_, t_) in pt} for k, m in enumerate(mols)}
        cia_t = [75.0, 200.0, 500.0, 1000.0, 2000.0, 4000.0]
        continuum = {pr: {t_: 10.0 ** (-7.0 + np.cos(wno / 4000.0 + k) + 0.3 * np.log10(t_ / 300.0)) for t_ in cia_t}
                     for k, pr in enumerate(("H2H2", "H2He"))}
        ray = {m: 1e-27 * (wno / 1e4) ** 4 for m in ("H2", "He")}
        t0 = time.perf_counter()
        opa = px.RetrieveOpacities(wno, pt, molecular, continuum, cia_t, rayleigh_opa=ray, query_method="linear", ctx=ctx)
        t_tables = time.perf_counter() - t0
        nlevel = 91
        plev = np.logspace(-6, 2, nlevel)
        prof = {"pressure": plev, "temperature": 150.0 + 1200.0 * ((np.log10(plev) + 6) / 8) ** 2,
                "H2": np.full(nlevel, 0.84), "He": np.full(nlevel, 0.155), "H2O": np.full(nlevel, 1e-3),
                "CH4": np.full(nlevel, 5e-4), "CO": np.full(nlevel, 1e-4), "NH3": np.full(nlevel, 1e-5)}
        opd = np.zeros((nlevel - 1, nwno)); opd[50:60] = 0.3
        w0c = np.zeros_like(opd); w0c[50:60] = 0.9
        g0c = np.zeros_like(opd); g0c[50:60] = 0.6
        case = jdi.inputs()
        case.phase_angle(0)
        case.gravity(gravity=2500.0)
        case.atmosphere(df=prof)
        case.clouds(df={"opd": opd, "w0": w0c, "g0": g0c})
        case.approx(raman="none")
        case.spectrum(opa, calculation="reflected+thermal")
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            case.spectrum(opa, calculation="reflected+thermal")
            ts.append(time.perf_counter() - t0)
        clear = jdi.inputs()
        clear.phase_angle(0)
        clear.gravity(gravity=2500.0)
        clear.atmosphere(df=prof)
        clear.approx(raman="none")
        clear.spectrum(opa, calculation="reflected+thermal")
        tc = []
        for _ in range(5):
            t0 = time.perf_counter()
            clear.spectrum(opa, calculation="reflected+thermal")
            tc.append(time.perf_counter() - t0)
        out["spectrum_e2e_1e5_cloud_free"] = dict(spectrum_s=min(tc))
        # BASELINE configs[4] per-GPU shard: 8 x 8 facets x 12 500 wavelengths x 90 layers, per-facet T
        nw3 = 12500
        opa3 = px.RetrieveOpacities(wno[::8], pt, {m: {i: v[::8] for i, v in d_.items()} for m, d_ in molecular.items()},
                                    {p_: {t_: v[::8] for t_, v in d_.items()} for p_, d_ in continuum.items()}, cia_t,
                                    rayleigh_opa={m: v[::8] for m, v in ray.items()}, query_method="linear", ctx=ctx)
        ng3 = nt3 = 8
        prof3 = {k: v for k, v in prof.items()}
        pert = 1.0 + 0.1 * np.cos(np.arange(64).reshape(8, 8))
        prof3["temperature"] = prof["temperature"][:, None, None] * pert[None]
        c3 = jdi.inputs()
        c3.phase_angle(np.pi / 3, num_gangle=ng3, num_tangle=nt3)
        c3.gravity(gravity=2500.0)
        c3.atmosphere_3d(prof3)
        c3.approx(raman="none")
        c3.spectrum(opa3, calculation="reflected+thermal", dimension="3d")
        t3 = []
        for _ in range(3):
            t0 = time.perf_counter()
            c3.spectrum(opa3, calculation="reflected+thermal", dimension="3d")
            t3.append(time.perf_counter() - t0)
        out["spectrum_3d_e2e_8x8x%d" % opa3.nwno] = dict(spectrum_s=min(t3))
        out["spectrum_e2e_1e5"] = dict(table_upload_s=t_tables, spectrum_s=min(ts),
                                       note="inputs.spectrum(reflected+thermal): host set-up, cloud planes H2D "
                                            "(3 x 72 MB), opacity interpolation + mixing + both solvers on the GPU")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
