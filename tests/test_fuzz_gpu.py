"""Randomised differential test: the HIP library against the CPU oracle over seeded random
combinations of sizes (including ragged last blocks and one or two layers), geometry (1-D Gauss
tables, ng x nt grids at a phase angle), phase-function / coefficient options, delta-Eddington,
surface reflectivity, stellar flux and cloud structure.  Small scenes, so the oracle takes
milliseconds; the point is coverage of option combinations and wave-uniform fast paths
(cloud-free layers, cumulative-tau planes, symmetric geometry) that the fixtures hit only singly.
Entries below 1e-4 of a field's maximum (limb facets, where exp(-tau/mu) all but vanishes) are judged
against that scale: they carry the rounding of O(max) terms in the reference as well."""
import os

import numpy as np
import pytest

from helpers import PLANES, excess, lvl_err, lvl_excess, rel_err

TTHG = (1.0, -1.0, 2.0, -0.5, 1.0)
# PICASO_FUZZ_OFFSET=<int> shifts every seed: a soak run walks through fresh combinations
OFFSET = int(__import__("os").environ.get("PICASO_FUZZ_OFFSET", "0"))


LOOSE = float(__import__("os").environ.get("PICASO_FUZZ_LOOSE", "3e-7"))


def _tol(sc, tight, loose):
    """The reference's formulas carry exp(+lambda*dtau) terms (clipped at 35) and lose about
    exp(lambda*dtau) * eps of relative precision in a layer of optical depth dtau; two evaluations of
    them in different operation order differ by that much.  Scenes are therefore held to
    `tight` (1e-8) while every layer is thin, to 2e-15 * exp(2 dtau_max) in between, and to `loose` once a layer
    reaches the clip.  `loose` for the top-of-atmosphere results is LOOSE = 3e-7 (PICASO_FUZZ_LOOSE overrides): 1.5 x the
    largest distance from the extended-precision answer seen in ~260 000 combinations (round 3 soak, single layers of
    optical depth 44-102), a third of BASELINE's 1e-6 contract -- until round 3 it WAS the contract, so a 1e-7
    regression on thick layers could pass."""
    worst = float(np.max(sc["dtau_og"]))
    return float(np.clip(2e-15 * np.exp(min(2.0 * worst, 35.0)), tight, loose))


def _dump(name, tag, **arrays):
    """PICASO_FUZZ_DUMP=dir: inputs and both results of a case, for a look at it off the GPU box
    (tools/fuzz_case_x80.py evaluates the reference in extended precision on them)."""
    d = os.environ.get("PICASO_FUZZ_DUMP")
    if d:
        os.makedirs(d, exist_ok=True)
        np.savez(os.path.join(d, "%s_off%d_%s.npz" % (name, OFFSET, "_".join(str(int(t)) for t in tag[:2]))),
                 tag=np.array([float(np.sum(t)) if np.ndim(t) else float(t) for t in tag]), **arrays)


def _geometry(rng):
    from picaso_amd import disco
    if rng.random() < 0.5:
        ng = int(rng.choice([5, 6, 7, 8]))
        g, gw, t, tw = disco.get_angles_1d(ng)
        u0, u1, ct, _, _ = disco.compute_disco(ng, 1, g, t, 0.0)
        return ng, 1, gw, tw, u0, u1, 1.0
    ng, nt = int(rng.integers(2, 6)), int(rng.integers(2, 5))
    g, gw, t, tw = disco.get_angles_3d(ng, nt)
    u0, u1, ct, _, _ = disco.compute_disco(ng, nt, g, t, float(rng.choice([0.0, 0.4, 1.1, 2.0])))
    return ng, nt, gw, tw, u0, u1, ct


def _scene(rng, seed):
    from picaso_amd import synthetic as syn
    nlayer = int(rng.choice([1, 2, 3, 7, 19, 40]))
    nwno = int(rng.choice([1, 5, 63, 64, 65, 130, 257]))
    kind = rng.integers(0, 4)
    kw = dict(seed=int(seed), delta_eddington=bool(rng.integers(0, 2)))
    if kind == 0:
        kw.update(cloud=False)
    elif kind == 1:
        kw.update(cloud_opd=float(10.0 ** rng.uniform(-2, 1.5)))
    elif kind == 2:
        kw.update(gas_scale=float(10.0 ** rng.uniform(-3, 2)), ray_scale=float(10.0 ** rng.uniform(-1, 1)))
    return syn.make_scene(nlayer, nwno, **kw), nlayer, nwno


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(6))
def test_fuzz_reflected(oracle, block):
    from picaso_amd import fluxes
    rng = np.random.default_rng(1000 + block + 7919 * OFFSET)
    for it in range(20):
        sc, nlayer, nwno = _scene(rng, 50 * block + it)
        ng, nt, gw, tw, u0, u1, ct = _geometry(rng)
        sp, mp, tc = int(rng.integers(0, 4)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
        lvl = int(rng.integers(0, 2))
        rs = float(rng.choice([0.0, 0.3, 1.0])) if rng.random() < 0.7 else rng.random(nwno)
        f0 = np.ones(nwno) if rng.random() < 0.5 else 0.2 + rng.random(nwno)
        b_top = float(rng.choice([0.0, 0.0, 0.5]))
        args = (nlayer + 1, sc["wno"], nwno, ng, nt, *[sc[k] for k in PLANES], rs, u0, u1, ct, f0, sp, mp, *TTHG)
        kw = dict(get_toa_intensity=1, get_lvl_flux=lvl, toon_coefficients=tc, b_top=b_top)
        xg, lg = fluxes.get_reflected_1d(*args, **kw)
        xo, lo = oracle.get_reflected_1d(*args, **kw)
        tag = (block, it, nlayer, nwno, ng, nt, sp, mp, tc, lvl)
        # intensities far below the incident flux (albedo < 1e-6: a nearly black layer) are what is left of
        # O(F0PI) terms cancelling; they are judged against 1e-6 of the incident flux
        floor = max(1e-4 * np.abs(xo).max(), 1e-6 * float(np.max(f0)))
        x80 = None
        if not rel_err(xg, xo, floor) < _tol(sc, 1e-8, LOOSE):
            # beyond the tolerance: is it the reference's own fp64 rounding?  (see the level fluxes below)
            x80 = oracle.get_reflected_1d(*args, x80=True, **kw)
            assert excess(xg, xo, x80[0], _tol(sc, 1e-8, LOOSE), floor) <= 0.0, tag
        if lvl:
            # contract tolerance: with single layers of optical depth 50-2000 (these random scenes have
            # them) the reference's level-flux expressions combine exp(+35)-sized terms and the upward
            # flux differs by up to 2e-7 of the field scale between formulations (median 8e-11).  In about one
            # scene in 400 the difference is larger (up to 1e-4 seen): every time it is the reference's fp64
            # rounding -- its formulas evaluated in x87 extended precision (the oracle's `x80=True` build of the
            # same source) move by exactly that much, and the kernel sits 100-1000 x closer to the extended
            # value than the fp64 reference does (tools/fuzz_case_x80.py on the reference itself).
            # Such a scene is held to helpers.lvl_excess: within tol + 2 e_ref of the reference, within
            # tol + e_ref/500 of the extended-precision value, element by element.
            if not lvl_err(lg, lo) < _tol(sc, 1e-6, 1e-5):
                _dump("reflected_lvl", tag, u0=u0, u1=u1, ct=ct, rs=rs, f0=f0, b_top=b_top, opts=np.array([sp, mp, tc]),
                      wno=sc["wno"], lg=np.stack(lg), lo=np.stack(lo), xg=xg, xo=xo, **{k: sc[k] for k in PLANES})
                x80 = x80 if x80 is not None else oracle.get_reflected_1d(*args, x80=True, **kw)
                assert lvl_excess(lg, lo, x80[1], _tol(sc, 1e-6, 1e-5)) <= 0.0, tag
        ag = oracle.compress_disco(nwno, ct, xo, gw, tw, f0)
        from picaso_amd import disco
        alb, afloor = disco.compress_disco(nwno, ct, xg, gw, tw, f0), max(1e-4 * np.abs(ag).max(), 1e-6)
        if not rel_err(alb, ag, afloor) < _tol(sc, 1e-8, LOOSE):
            x80 = x80 if x80 is not None else oracle.get_reflected_1d(*args, x80=True, **kw)
            assert excess(alb, ag, oracle.compress_disco(nwno, ct, x80[0], gw, tw, f0), _tol(sc, 1e-8, LOOSE),
                          afloor) <= 0.0, tag


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(4))
def test_fuzz_thermal(oracle, block):
    from picaso_amd import fluxes
    rng = np.random.default_rng(2000 + block + 7919 * OFFSET)
    for it in range(20):
        sc, nlayer, nwno = _scene(rng, 900 + 50 * block + it)
        ng, nt, gw, tw, u0, u1, ct = _geometry(rng)
        hard = int(rng.integers(0, 2))
        calc = int(rng.integers(0, 2))
        rs = float(rng.choice([0.0, 0.2])) if rng.random() < 0.7 else 0.5 * rng.random(nwno)
        dw = np.abs(np.gradient(sc["wno"])) if nwno > 1 else np.array([10.0])
        args = (nlayer + 1, sc["wno"], nwno, ng, nt, sc["tlevel"], sc["dtau_og"], sc["w0_no_raman"], sc["cosb_og"],
                sc["plevel"], u1, rs, hard, dw, calc)
        fg, _ = fluxes.get_thermal_1d(*args)
        fo, _ = oracle.get_thermal_1d(*args)
        if not rel_err(fg, fo, 1e-4 * np.abs(fo).max()) < _tol(sc, 1e-8, LOOSE):
            _dump("thermal", (block, it), u1=u1, rs=rs, hard=hard, dw=dw, calc=calc, fg=fg, fo=fo,
                  **{k: sc[k] for k in ("wno", "tlevel", "plevel", "dtau_og", "w0_no_raman", "cosb_og")})
            f80, _ = oracle.get_thermal_1d(*args, x80=True)      # the reference's own fp64 rounding? (see above)
            assert excess(fg, fo, f80, _tol(sc, 1e-8, LOOSE), 1e-4 * np.abs(fo).max()) <= 0.0, (block, it, nlayer, nwno,
                                                                                                ng, nt, hard, calc)


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(3))
def test_fuzz_spherical_harmonics(oracle, block):
    from picaso_amd import fluxes
    rng = np.random.default_rng(3000 + block + 7919 * OFFSET)
    for it in range(12):
        sc, nlayer, nwno = _scene(rng, 1800 + 50 * block + it)
        stream = int(rng.choice([2, 4]))
        # the SH planes carry their own delta-M factor: rebuild the scene with the matching stream
        from picaso_amd import synthetic as syn
        sc = syn.make_scene(nlayer, nwno, seed=1800 + 50 * block + it, stream=stream)
        ng, nt, gw, tw, u0, u1, ct = _geometry(rng)
        opts = [int(rng.integers(0, 3)) for _ in range(3)] + [int(rng.integers(0, 2)) for _ in range(3)]   # forms: TTHG / OTHG / isotropic
        sform = int(rng.integers(0, 2))
        rs = float(rng.choice([0.0, 0.3]))
        args = lambda fd: (nlayer + 1, nwno, ng, nt, sc["dtau"], sc["tau"], sc["w0"], sc["cosb"], sc["ftau_cld"],
                           sc["ftau_ray"], fd, sc["dtau_og"], sc["tau_og"], sc["w0_og"], sc["cosb_og"], rs, u0, u1,
                           ct, np.ones(nwno), *opts, *TTHG, stream)
        xg, _ = fluxes.get_reflected_SH(*args(sc["f_deltaM"].copy()), b_top=0.0, flx=0, single_form=sform)
        xo, _ = oracle.get_reflected_SH(*args(sc["f_deltaM"].copy()), b_top=0.0, flx=0, single_form=sform)
        # observed max over the four blocks (round 5): 1.3e-12; one draw of offsets 1-300 at 2.3e-8 (see _sh_close)
        _sh_close(oracle, xg, xo, args(sc["f_deltaM"].copy()), dict(b_top=0.0, flx=0, single_form=sform),
                  float(sc["w0"].max()), (block, it, nlayer, nwno, ng, nt, stream, opts, sform))


def _sh_close(oracle, xg, xo, args, kwargs, w0max, tag, tol=1e-9, loose=1e-7, cap=3e-7, factor=30.0):
    """``xg`` (kernel) against ``xo`` (the fp64 oracle) at ``tol``.  Where that fails the column set must be one the
    reference's OWN formulas are ill-conditioned on, and the measure of that is the reference itself: the distance of the
    fp64 oracle from the x87 extended-precision evaluation of the same restatement (oracle/sh_oracle_x80.c).  The kernel is
    then held to ``factor`` x that distance against the x87 value (at least ``tol``; at least ``loose`` where max w0 > 0.999,
    round 5's rule), never more than ``cap`` -- the cap of the Toon draws, a third of BASELINE's 1e-6.
    Round 5 (offsets 0-2 300) knew one such family -- nearly conservative scattering, w0 > 0.999, where the SH4 modes of
    fluxes.py:3388-3434 lose digits: 17 of ~11 000 draws beyond 1e-9, kernel vs x87 at most 2.5e-8, 1.5 - 22 x the oracle's own
    distance.  Round 6 (offsets 2 307-2 506, 4 of ~7 000 draws) met it at max w0 = 0.9987 and 0.9978 (kernel CLOSER to x87 than
    the oracle in both), once at w0 = 0.99999 with 1.9e-7 (oracle 5.8e-8), and a second family at w0 = 0.946: 1/ubar0 next to an
    SH4 eigenvalue, the singularity of the beam's particular solution (fluxes.py:3397-3416; oracle 1.1e-9 from x87, kernel 7.4e-9).
    Hence the oracle's own distance instead of a w0 threshold."""
    floor = 1e-4 * np.abs(xo).max()
    if rel_err(xg, xo, floor) < tol:
        return
    xx, _ = oracle.get_reflected_SH(*args, **kwargs, x80=True)
    e_ref, e_k = rel_err(xo, xx, floor), rel_err(xg, xx, floor)
    allowed = min(cap, max(tol, factor * e_ref, loose if w0max > 0.999 else 0.0))
    assert e_k < allowed, (tag, "max w0 %.6f, kernel vs x87 %.2e, fp64 oracle vs x87 %.2e, allowed %.2e"
                           % (w0max, e_k, e_ref, allowed))


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(4))
def test_fuzz_sh4_cloud_free_form(oracle, block):
    """get_reflected_SH handed dtau and w0 only (k_sh4_clear) against the oracle on all eleven planes of the same
    cloud-free scene: random sizes, geometry, optical thickness, surface, stellar flux and top boundary."""
    from picaso_amd import fluxes
    from picaso_amd import synthetic as syn
    rng = np.random.default_rng(3500 + block + 7919 * OFFSET)
    for it in range(12):
        nlayer = int(rng.choice([1, 2, 3, 7, 19, 40, 90]))
        nwno = int(rng.choice([1, 5, 63, 64, 65, 130, 257, 300]))
        sc = syn.make_scene(nlayer, nwno, seed=2600 + 50 * block + it, stream=4, cloud=False,
                            gas_scale=float(10.0 ** rng.uniform(-4, 2.5)), ray_scale=float(10.0 ** rng.uniform(-2, 2)))
        ng, nt, gw, tw, u0, u1, ct = _geometry(rng)
        rs = rng.random(nwno) * float(rng.choice([0.0, 0.3, 1.0]))
        f0 = 1.0 + rng.random(nwno) if rng.random() < 0.5 else np.ones(nwno)
        b_top = float(rng.choice([0.0, 0.2]))
        tail = (rs, u0, u1, ct, f0, 0, 0, 0, 1, 1, 1, *TTHG, 4, b_top)
        full = [sc[k] for k in ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray", "f_deltaM", "dtau_og", "tau_og", "w0_og",
                                "cosb_og")]
        lean = [sc["dtau"], None, sc["w0"]] + [None] * 8
        xg, _ = fluxes.get_reflected_SH(nlayer + 1, nwno, ng, nt, *lean, *tail)
        xo, _ = oracle.get_reflected_SH(nlayer + 1, nwno, ng, nt, *[np.array(a) for a in full], *tail)
        assert np.isfinite(xg).all()
        # observed max over the four blocks (round 5): 3.4e-11; nearly conservative draws of the soak: see _sh_close
        _sh_close(oracle, xg, xo, (nlayer + 1, nwno, ng, nt, *[np.array(a) for a in full], *tail), {},
                  float(sc["w0"].max()), (block, it, nlayer, nwno, ng, nt, b_top))


def _facet_planes(rng, nlayer, nwno, ng, nt, seed):
    """Per-facet planes (rows, nwno, ng, nt): every facet its own random scene."""
    from picaso_amd import synthetic as syn
    scs = [[syn.make_scene(nlayer, nwno, seed=seed + 31 * (g * nt + t), cloud=bool((g + t) % 2),
                           gas_scale=float(10.0 ** rng.uniform(-2, 1))) for t in range(nt)] for g in range(ng)]
    keys = PLANES + ("w0_no_raman",)
    st = {k: np.ascontiguousarray(np.stack([np.stack([scs[g][t][k] for t in range(nt)], axis=2)
                                            for g in range(ng)], axis=2)) for k in keys}
    tl = np.stack([np.stack([scs[g][t]["tlevel"] * (1 + 0.05 * g - 0.03 * t) for t in range(nt)], axis=1)
                   for g in range(ng)], axis=1)
    pl = np.stack([np.stack([scs[g][t]["plevel"] for t in range(nt)], axis=1) for g in range(ng)], axis=1)
    return scs[0][0], st, tl, pl


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(3))
def test_fuzz_facets_3d(oracle, block):
    from picaso_amd import disco, fluxes
    rng = np.random.default_rng(4000 + block + 7919 * OFFSET)
    for it in range(6):
        nlayer = int(rng.choice([1, 4, 17]))
        nwno = int(rng.choice([3, 33, 70]))
        ng, nt = int(rng.integers(2, 6)), int(rng.integers(2, 5))
        g, gw, t, tw = disco.get_angles_3d(ng, nt)
        u0, u1, ct, _, _ = disco.compute_disco(ng, nt, g, t, float(rng.choice([0.0, 0.7, 1.6])))
        sc0, st, tl, pl = _facet_planes(rng, nlayer, nwno, ng, nt, 7000 + 100 * block + it)
        sp, mp = int(rng.integers(0, 4)), int(rng.integers(0, 2))
        rs = float(rng.choice([0.0, 0.4]))
        a = (nlayer + 1, sc0["wno"], nwno, ng, nt, *[st[k] for k in PLANES], rs, u0, u1, ct, np.ones(nwno), sp, mp,
             *TTHG)
        xg, xo = fluxes.get_reflected_3d(*a), oracle.get_reflected_3d(*a)
        assert rel_err(xg, xo, 1e-4 * np.abs(xo).max()) < _tol(st, 1e-8, LOOSE), (block, it, nlayer, nwno, ng, nt, sp, mp)
        hs = int(rng.integers(0, 2))
        b = (nlayer + 1, sc0["wno"], nwno, ng, nt, tl, st["dtau_og"], st["w0_no_raman"], st["cosb_og"], pl, u1, rs, hs)
        fg, fo = fluxes.get_thermal_3d(*b), oracle.get_thermal_3d(*b)
        assert rel_err(fg, fo, 1e-4 * np.abs(fo).max()) < _tol(st, 1e-8, LOOSE), (block, it, nlayer, nwno, ng, nt, hs)


@pytest.mark.gpu
def test_fuzz_compute_opacity(oracle):
    """Mixing + delta-Eddington kernel against the numpy restatement on random optical depths,
    all test modes / streams, with and without a Raman plane and clouds."""
    from oracle import optics_oracle as oo
    from picaso_amd import _lib
    from picaso_amd._lib import check, load, ptr
    from picaso_amd.device import DeviceArray
    import ctypes
    names = ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray", "gcos2", "dtau_og", "tau_og", "w0_og", "cosb_og",
             "w0_no_raman", "f_deltaM")
    ctx = _lib.context()
    rng = np.random.default_rng(55 + 7919 * OFFSET)
    for it in range(24):
        nlayer, nwno = int(rng.choice([1, 2, 9, 31])), int(rng.choice([1, 17, 64, 300]))
        tg = 10.0 ** rng.uniform(-6, 2, (nlayer, nwno))
        tr = 10.0 ** rng.uniform(-6, 1, (nlayer, nwno))
        cloudy = it % 3 != 0
        tc = np.where(rng.random((nlayer, nwno)) < 0.4, 10.0 ** rng.uniform(-3, 1, (nlayer, nwno)), 0.0) * cloudy
        wc, gc = 0.2 + 0.79 * rng.random((nlayer, nwno)), 0.95 * rng.random((nlayer, nwno))
        null_cloud = (not cloudy) and it % 2 == 0            # NULL cloud planes = zero planes
        if null_cloud:
            wc, gc = np.zeros_like(wc), np.zeros_like(gc)
        raman = rng.random((nlayer, nwno)) * 0.99999 if it % 2 else None
        tm = [None, "rayleigh", "constant_tau"][it % 3] if it % 4 == 0 else None
        de, stream = bool(rng.integers(0, 2)), int(rng.choice([2, 4]))
        want = oo.compute_opacity(tg, tr, tc, wc, gc, raman if raman is not None else 0.99999, stream=stream,
                                  delta_eddington=de, test_mode=tm)
        d = {k: DeviceArray.from_host(v, ctx) for k, v in dict(tg=tg, tr=tr, tc=tc, wc=wc, gc=gc).items()}
        d_r = DeviceArray.from_host(raman, ctx) if raman is not None else None
        outs = [DeviceArray((nlayer + 1 if k in ("tau", "tau_og") else nlayer, nwno), ctx) for k in names]
        check(load().picaso_compute_opacity_dev(
            ctx, ctypes.c_int(nlayer), ctypes.c_int(nwno), ptr(d["tg"].addr), ptr(d["tr"].addr),
            None if null_cloud else ptr(d["tc"].addr), None if null_cloud else ptr(d["wc"].addr),
            None if null_cloud else ptr(d["gc"].addr), ptr(d_r.addr) if d_r else None, ctypes.c_double(0.99999),
            ctypes.c_int({None: 0, "rayleigh": 1, "constant_tau": 2}[tm]), ctypes.c_int(int(de)),
            ctypes.c_int(stream), *[ptr(o.addr) for o in outs]), ctx)
        for k, o, w in zip(names, outs, want):
            got = o.to_host()
            w = np.broadcast_to(w, got.shape)
            assert rel_err(got, w, 1e-300) < 1e-12, (it, k, nlayer, nwno, tm, de, stream)


@pytest.mark.gpu
def test_fuzz_fused_opacity_equals_two_launches():
    """picaso_gas_compute_opacity_dev (gas stage + mixing in ONE launch, level sums apart) against
    picaso_opacity_gas_ck_dev -> picaso_compute_opacity_ck_dev on random tables, table rows, weights and coefficients,
    with and without cloud / Raman planes, every test mode, and random subsets of the thirteen outputs (the
    cloud-free subset takes the specialised flavour of the kernel): np.array_equal plane by plane."""
    import ctypes
    from picaso_amd import _lib
    from picaso_amd._lib import check, load, ptr
    from picaso_amd.device import DeviceArray
    names = ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray", "gcos2", "dtau_og", "tau_og", "w0_og", "cosb_og",
             "w0_no_raman", "f_deltaM")
    ci, cd = ctypes.c_int, ctypes.c_double
    ip = ctypes.POINTER(ctypes.c_int)
    dpp = ctypes.POINTER(ctypes.POINTER(ctypes.c_double))
    ctx = _lib.context()
    rng = np.random.default_rng(901 + 7919 * OFFSET)
    for it in range(30):
        nlayer, nwno = int(rng.choice([1, 5, 6, 7, 13, 31])), int(rng.choice([1, 17, 256, 300]))
        nmol, ncont, nray, nrows = int(rng.integers(0, 4)), int(rng.integers(0, 3)), int(rng.integers(1, 3)), 7
        mol_mode = int(rng.integers(0, 2))
        tabs = [DeviceArray.from_host(rng.uniform(-26, -20, (nrows, nwno)) if mol_mode else
                                      10.0 ** rng.uniform(-26, -20, (nrows, nwno)), ctx) for _ in range(nmol)]
        ctabs = [DeviceArray.from_host(10.0 ** rng.uniform(-8, -5, (nrows, nwno)), ctx) for _ in range(ncont)]
        rtabs = [DeviceArray.from_host(10.0 ** rng.uniform(-28, -25, (nwno,)), ctx) for _ in range(nray)]

        def parr(xs):
            a = (ctypes.c_void_p * max(1, len(xs)))(*[x.addr for x in xs])
            return a, ctypes.cast(a, dpp)
        k1, p1 = parr(tabs)
        k2, p2 = parr(ctabs)
        k3, p3 = parr(rtabs)
        rows = np.ascontiguousarray(rng.integers(0, nrows, (max(nmol, 1), nlayer, 4)), dtype=np.int32)
        wts = np.ascontiguousarray(rng.random((max(nmol, 1), nlayer, 4)))
        mfac = np.ascontiguousarray(10.0 ** rng.uniform(-3, 1, (max(nmol, 1), nlayer)))
        crows = np.ascontiguousarray(rng.integers(0, nrows, (max(ncont, 1), nlayer)), dtype=np.int32)
        cfac = np.ascontiguousarray(10.0 ** rng.uniform(2, 6, (max(ncont, 1), nlayer)))
        rfac = np.ascontiguousarray(10.0 ** rng.uniform(22, 26, (nray, nlayer)))
        gas = (ci(mol_mode), ci(nmol), p1, rows.ctypes.data_as(ip), ptr(wts), ptr(mfac), ci(0), ci(ncont), p2,
               crows.ctypes.data_as(ip), None, ptr(cfac), ci(nray), p3, ptr(rfac))
        cloudy = it % 3 != 0
        cl = [DeviceArray.from_host(x, ctx) for x in (
            np.where(rng.random((nlayer, nwno)) < 0.4, 10.0 ** rng.uniform(-3, 1, (nlayer, nwno)), 0.0),
            0.2 + 0.79 * rng.random((nlayer, nwno)), 0.95 * rng.random((nlayer, nwno)))] if cloudy else [None] * 3
        # every other cloudy case: the cloud as tables on a grid of their own -- regridded planes (picaso_regrid_rows_dev)
        # for the two launches, the tables themselves for the fused one
        tab = (0, None, None, None)
        if cloudy and it % 2:
            from picaso_amd import device as pdev
            nin = int(rng.integers(2, 40))
            xg = np.sort(rng.uniform(900.0, 36000.0, nin))
            xw = np.sort(rng.uniform(1000.0, 35000.0, nwno))
            if nwno > 2:
                xw[1] = xg[min(1, nin - 1)]                          # a knot, and points outside the table's grid
                xw[0], xw[-1] = xg[0] - 5.0, xg[-1] + 5.0
            fp = np.concatenate([10.0 ** rng.uniform(-3, 1, (nlayer, nin)) * (rng.random((nlayer, nin)) < 0.6),
                                 0.2 + 0.79 * rng.random((nlayer, nin)), 0.95 * rng.random((nlayer, nin))])
            d_x = DeviceArray.from_host(xw, ctx)
            planes = pdev.regrid_rows(xg, fp, d_x, ctx).reshape((3, nlayer, nwno))
            cl = [planes.row_block(0), planes.row_block(1), planes.row_block(2)]
            d_xg, d_fp = DeviceArray.from_host(xg, ctx), DeviceArray.from_host(fp, ctx)
            tab = (nin, d_xg, d_fp, d_x)
        rmode = it % 3                                   # Raman: none, a plane, one row
        raman = None if rmode == 0 else DeviceArray.from_host(
            rng.random((nlayer, nwno) if rmode == 1 else (nwno,)) * 0.99999, ctx)
        tm = int(rng.integers(0, 3)) if (cloudy and it % 4 == 0) else 0
        de, stream = int(rng.integers(0, 2)), int(rng.choice([2, 4]))
        if it % 5 == 0:
            want = set(names)
        elif not cloudy and tm == 0:
            want = {"dtau", "w0"} | ({"tau"} if it % 2 else set()) | ({"w0_no_raman"} if it % 4 == 1 else set())
        else:
            want = {k for k in names if rng.random() < 0.6} | {"dtau", "dtau_og"}
        mix = (*[ptr(x.addr) if x is not None else None for x in cl], ptr(raman.addr) if raman is not None else None,
               ci(nlayer if rmode == 1 else 0), cd(0.99999), ci(tm), ci(de), ci(stream))

        def outs():
            return [DeviceArray.zeros((nlayer + 1 if k in ("tau", "tau_og") else nlayer, nwno), ctx) if k in want else None
                    for k in names]
        o1, o2 = outs(), outs()
        tg, tr = DeviceArray((nlayer, nwno), ctx), DeviceArray((nlayer, nwno), ctx)
        check(load().picaso_opacity_gas_ck_dev(ctx, ci(nlayer), ci(nwno), ci(1), *gas, ptr(tg.addr), ptr(tr.addr)), ctx)
        check(load().picaso_compute_opacity_ck_dev(ctx, ci(nlayer), ci(nwno), ci(1), ptr(tg.addr), ptr(tr.addr), *mix,
                                                   *[ptr(o.addr) if o is not None else None for o in o1]), ctx)
        mix2 = mix if not tab[0] else (None, None, None) + mix[3:]
        check(load().picaso_gas_compute_opacity_dev(ctx, ci(nlayer), ci(nwno), *gas, *mix2,
                                                    *[ptr(o.addr) if o is not None else None for o in o2],
                                                    ci(1), ci(tab[0]), *[ptr(x.addr) if x is not None else None for x in tab[1:]]),
              ctx)
        for k, a, b in zip(names, o1, o2):
            if a is not None:
                ha, hb = a.to_host(), b.to_host()
                assert np.all(np.isfinite(ha)) or tm, (it, k)
                assert np.array_equal(ha, hb, equal_nan=True), (it, k, nlayer, nwno, nmol, ncont, cloudy, rmode, tm, de, sorted(want))
