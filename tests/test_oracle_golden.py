"""Pin the CPU oracle (oracle/picaso_oracle.c) against golden vectors produced by the reference's
own source (tests/golden/make_golden.py).  Tolerance: 1e-11 relative on intensities/fluxes at the
top of the atmosphere (observed ~1e-13: libm-vs-numpy last-ulp differences only); level fluxes are
judged against the per-wavelength field scale because the reference's own downward-flux expressions
cancel catastrophically in thin layers."""
import numpy as np
import pytest

from helpers import PLANES, Golden, golden_files, lvl_err, rel_err, scale_err, scene_id

TOL = 1e-11
FILES_1D = golden_files("scene1d_")
FILES_3D = golden_files("scene3d_")


@pytest.mark.parametrize("path", FILES_1D, ids=scene_id)
def test_reflected_1d(path, oracle):
    g = Golden(path)
    nlevel, nwno = g.inp("tau").shape
    planes = [g.inp(k) for k in PLANES]
    for case in g.cases("refl1d"):
        sp, mp, tc, lvl = (int(s[-1]) for s in case.split("_"))
        b_top = float(g["refl1d/%s/b_top" % case])
        xint, lv = oracle.get_reflected_1d(
            nlevel, g.inp("wno"), nwno, g.geo("numg"), g.geo("numt"), *planes,
            g.inp("surf_reflect"), g.geo("ubar0"), g.geo("ubar1"), g.geo("cos_theta"),
            g.inp("F0PI"), sp, mp, *g.tthg(), get_toa_intensity=1, get_lvl_flux=lvl,
            toon_coefficients=tc, b_top=b_top)
        assert rel_err(xint, g["refl1d/%s/xint" % case]) < TOL, case
        if lvl:
            ref4 = [g["refl1d/%s/%s" % (case, nm)] for nm in ("fm", "fp", "fmm", "fpm")]
            assert lvl_err(lv, ref4) < TOL, case


@pytest.mark.parametrize("path", FILES_1D, ids=scene_id)
def test_thermal_1d(path, oracle):
    g = Golden(path)
    nlevel, nwno = g.inp("tau").shape
    for case in g.cases("therm1d"):
        hs, ct = (int(s[-1]) for s in case.split("_"))
        rs = np.zeros(nwno) + g.inp("surf_reflect")
        flux, lv = oracle.get_thermal_1d(nlevel, g.inp("wno"), nwno, g.geo("numg"), g.geo("numt"),
                                         g.inp("tlevel"), g.inp("dtau_og"), g.inp("w0_no_raman"),
                                         g.inp("cosb_og"), g.inp("plevel"), g.geo("ubar1"), rs, hs,
                                         g["dwno"], ct)
        assert rel_err(flux, g["therm1d/%s/flux" % case]) < TOL, case
        if "therm1d/%s/fm" % case in g.keys:
            ref4 = [g["therm1d/%s/%s" % (case, nm)] for nm in ("fm", "fp", "fmm", "fpm")]
            assert lvl_err(lv, ref4) < TOL, case


@pytest.mark.parametrize("path", FILES_1D, ids=scene_id)
def test_thermal_1d_extended_precision_build(path, oracle):
    """The `x80=True` build of the restatement (same source, real = long double) against the reference's own
    level fluxes evaluated in numpy longdouble (the `_x80` arrays of the fixtures, tests/golden/make_golden.py
    `_extended`): both are x87 extended evaluations of the same expressions, stored rounded to float64 -- they
    agree to fp64 rounding of the field scale, on scenes where the reference's fp64 level fluxes are off by up to 1.5e-4
    of the field scale (thick layers: 7.4e-6 on cfg3like, 1.5e-4 on jupiterlike)."""
    g = Golden(path)
    nlevel, nwno = g.inp("tau").shape
    for case in g.cases("therm1d"):
        if "therm1d/%s/fm_x80" % case not in g.keys:
            continue
        hs, ct = (int(s[-1]) for s in case.split("_"))
        rs = np.zeros(nwno) + g.inp("surf_reflect")
        args = (nlevel, g.inp("wno"), nwno, g.geo("numg"), g.geo("numt"), g.inp("tlevel"), g.inp("dtau_og"),
                g.inp("w0_no_raman"), g.inp("cosb_og"), g.inp("plevel"), g.geo("ubar1"), rs, hs, g["dwno"], ct)
        _, lv80 = oracle.get_thermal_1d(*args, x80=True)
        want = [g["therm1d/%s/%s_x80" % (case, nm)] for nm in ("fm", "fp", "fmm", "fpm")]
        assert lvl_err(lv80, want) < 1e-14, case


@pytest.mark.parametrize("path", FILES_1D + FILES_3D, ids=scene_id)
def test_compress(path, oracle):
    g = Golden(path)
    nwno = g.inp("wno").shape[0]
    fam = "refl1d" if "scene1d" in path else "refl3d"
    key = "sp3_mp0_tc0_lvl0" if fam == "refl1d" else "sp0_mp0"
    alb = oracle.compress_disco(nwno, g.geo("cos_theta"), g["%s/%s/xint" % (fam, key)],
                                g.geo("gweight"), g.geo("tweight"), g.inp("F0PI"))
    assert rel_err(alb, g["compress_disco/albedo"]) < 1e-13
    tfam, tkey = ("therm1d", "hs0_ct0") if fam == "refl1d" else ("therm3d", "hs0")
    fl = oracle.compress_thermal(nwno, g["%s/%s/flux" % (tfam, tkey)], g.geo("gweight"),
                                 g.geo("tweight"))
    assert rel_err(fl, g["compress_thermal/flux"]) < 1e-13
    if fam == "refl1d":
        fl4 = oracle.compress_thermal(nwno, g["therm1d/hs0_ct0/fp"], g.geo("gweight"),
                                      g.geo("tweight"))
        assert rel_err(fl4, g["compress_thermal/lvl_fp"]) < 1e-13


@pytest.mark.parametrize("path", FILES_3D, ids=scene_id)
def test_reflected_thermal_3d(path, oracle):
    g = Golden(path)
    nlevel, nwno = g.inp("tau").shape[:2]
    planes = [g.inp(k) for k in PLANES]
    for case in g.cases("refl3d"):
        sp, mp = (int(s[-1]) for s in case.split("_"))
        xint = oracle.get_reflected_3d(nlevel, g.inp("wno"), nwno, g.geo("numg"), g.geo("numt"),
                                       *planes, g.inp("surf_reflect"), g.geo("ubar0"),
                                       g.geo("ubar1"), g.geo("cos_theta"), g.inp("F0PI"), sp, mp,
                                       *g.tthg())
        assert rel_err(xint, g["refl3d/%s/xint" % case]) < TOL, case
    for hs in (0, 1):
        flux = oracle.get_thermal_3d(nlevel, g.inp("wno"), nwno, g.geo("numg"), g.geo("numt"),
                                     g.inp("tlevel"), g.inp("dtau_og"), g.inp("w0_no_raman"),
                                     g.inp("cosb_og"), g.inp("plevel"), g.geo("ubar1"),
                                     g.inp("surf_reflect"), hs)
        assert rel_err(flux, g["therm3d/hs%d/flux" % hs]) < TOL, hs


FILES_SH = golden_files("scene_sh_")


def _sh_case(case):
    s, f, r, sf = case.split("_")
    return int(s[1]), [int(c) for c in f[1:]], [int(c) for c in r[1:]], int(sf[2])


@pytest.mark.parametrize("path", FILES_SH, ids=scene_id)
def test_spherical_harmonics(path, oracle):
    """SH2 / SH4 reflected + thermal (banded LU with partial pivoting restated from LAPACK dgbsv)
    against the reference (scipy.linalg.solve_banded).  Agreement is limited by the conditioning
    of the reference's 11-diagonal system (entries spanning e^-35 .. e^+35), not by the port."""
    g = Golden(path)
    nlevel, nwno = g.inp("tau").shape
    for case in g.cases("reflsh"):
        stream, (wsf, wmf, psf), (wsr, wmr, psr), sf = _sh_case(case)
        has_flux = ("reflsh/%s/flux" % case) in g.z.files
        xint, flux = oracle.get_reflected_SH(
            nlevel, nwno, g.geo("numg"), g.geo("numt"), g.inp("dtau"), g.inp("tau"), g.inp("w0"),
            g.inp("cosb"), g.inp("ftau_cld"), g.inp("ftau_ray"), g.inp("f_deltaM_s%d" % stream).copy(),
            g.inp("dtau_og"), g.inp("tau_og"), g.inp("w0_og"), g.inp("cosb_og"), g.inp("surf_reflect"),
            g.geo("ubar0"), g.geo("ubar1"), g.geo("cos_theta"), g.inp("F0PI"), wsf, wmf, psf, wsr, wmr,
            psr, *g.tthg(), stream, b_top=0.0, flx=1 if has_flux else 0, single_form=sf)
        # the restatement follows LAPACK's arithmetic: it sits at <= 3.1e-12 on the committed fixtures, so 1e-10
        # (not the 1e-8 the conditioning argument alone would allow) catches a regression in sh_oracle.c
        assert rel_err(xint, g["reflsh/%s/xint" % case]) < 1e-10, case
        if has_flux:        # layer moment fluxes (flx=1): field-scale metric, entries span many decades
            assert scale_err(flux, g["reflsh/%s/flux" % case]) < 1e-10, case
    for case in g.cases("thermsh"):
        stream, hs = int(case[1]), int(case[-1])
        rs = np.zeros(nwno) + g.inp("surf_reflect")
        xint, _ = oracle.get_thermal_SH(nlevel, g.inp("wno"), nwno, g.geo("numg"), g.geo("numt"),
                                        g.inp("tlevel"), g.inp("dtau"), g.inp("tau"), g.inp("w0"),
                                        g.inp("cosb"), g.inp("dtau_og"), g.inp("tau_og"), g.inp("w0_og"),
                                        g.inp("w0_no_raman"), g.inp("cosb_og"), g.inp("plevel"),
                                        g.geo("ubar1"), rs, stream, hs)
        assert rel_err(xint, g["thermsh/%s/xint" % case]) < 1e-10, case


# ---- round 5 fixtures: SH form index 2 / b_top, and the correlated-k loop around SH and 3-D ----
@pytest.mark.parametrize("name", ["cfg3like", "phase60"])
def test_spherical_harmonics_isotropic_form_and_b_top(name, oracle):
    """Form index 2 ('isotropic': the reference falls through with its `ones` weights, fluxes.py:2805-2855) on each of the
    three form arguments, and b_top != 0 (sh_extra_<name>.npz; inputs are those of scene_sh_<name>.npz)."""
    import os
    from helpers import GOLDEN
    g = Golden(os.path.join(GOLDEN, "scene_sh_%s.npz" % name))
    x = Golden(os.path.join(GOLDEN, "sh_extra_%s.npz" % name))
    nlevel, nwno = g.inp("tau").shape

    def run(stream, forms, rays, sf, b_top):
        return oracle.get_reflected_SH(
            nlevel, nwno, g.geo("numg"), g.geo("numt"), g.inp("dtau"), g.inp("tau"), g.inp("w0"), g.inp("cosb"),
            g.inp("ftau_cld"), g.inp("ftau_ray"), g.inp("f_deltaM_s%d" % stream).copy(), g.inp("dtau_og"), g.inp("tau_og"),
            g.inp("w0_og"), g.inp("cosb_og"), g.inp("surf_reflect"), g.geo("ubar0"), g.geo("ubar1"), g.geo("cos_theta"),
            g.inp("F0PI"), *forms, *rays, *g.tthg(), stream, b_top=b_top, flx=0, single_form=sf)[0]
    cases = x.cases("reflsh")
    assert len(cases) == 14 and any("_f222_" in c for c in cases)
    for case in cases:
        stream, forms, rays, sf = _sh_case(case)
        assert rel_err(run(stream, forms, rays, sf, 0.0), x["reflsh/%s/xint" % case]) < 1e-10, case
    for case in x.cases("btop"):
        stream, forms, rays, sf = _sh_case(case)
        b_top = float(x["btop/%s/b_top" % case])
        assert b_top != 0.0
        assert rel_err(run(stream, forms, rays, sf, b_top), x["btop/%s/xint" % case]) < 1e-10, case


def _ck_rt():
    import os
    from helpers import GOLDEN
    return (np.load(os.path.join(GOLDEN, "ck_rt.npz")), np.load(os.path.join(GOLDEN, "ck.npz")),
            np.load(os.path.join(GOLDEN, "optics.npz")))


CK_NAMES = ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray", "gcos2", "dtau_og", "tau_og", "w0_og", "cosb_og",
            "w0_no_raman", "f_deltaM")


def test_sh_inside_the_correlated_k_loop(oracle):
    """justdoit.py:256-307 / 364-380 with the oracle in the solver's place: one SH solve per Gauss point on the
    slice plane[:, :, ig], accumulated with gauss_wts in ig order."""
    r, ck, og = _ck_rt()
    wno, wts = og["in/wno"], ck["in/gauss_wts"]
    nwno, nlevel = wno.size, r["sh/tlevel"].size
    rs, f0 = r["sh/surf_reflect"], r["sh/F0PI"]
    tthg = (1.0, -1.0, 2.0, -0.5, 1.0)
    n = 0
    for gname in ("g5", "g3x2"):
        geo = {k: r["sh/%s/geo/%s" % (gname, k)] for k in ("numg", "numt", "ubar0", "ubar1", "cos_theta", "gweight", "tweight")}
        ng, nt, ct = int(geo["numg"]), int(geo["numt"]), float(geo["cos_theta"])
        for stream in (2, 4):
            pl = {nm: ck["de1_s%d/%s" % (stream, nm)] for nm in CK_NAMES}
            sl = lambda nm, ig: np.ascontiguousarray(pl[nm][:, :, ig])
            for key in [k[:-len("/xint_at_top")] for k in r.files if k.startswith("sh/%s/s%d_" % (gname, stream))
                        and k.endswith("/xint_at_top")]:
                _, forms, rays, sf = _sh_case(key.split("/")[-1])
                acc = 0
                for ig in range(wts.size):
                    x, _ = oracle.get_reflected_SH(
                        nlevel, nwno, ng, nt, sl("dtau", ig), sl("tau", ig), sl("w0", ig), sl("cosb", ig), sl("ftau_cld", ig),
                        sl("ftau_ray", ig), sl("f_deltaM", ig), sl("dtau_og", ig), sl("tau_og", ig), sl("w0_og", ig),
                        sl("cosb_og", ig), rs, geo["ubar0"], geo["ubar1"], ct, f0, *forms, *rays, *tthg, stream,
                        b_top=0.0, flx=0, single_form=sf)
                    acc = acc + x * wts[ig]
                assert rel_err(acc, r[key + "/xint_at_top"]) < 1e-10, key
                alb = oracle.compress_disco(nwno, ct, acc, geo["gweight"], geo["tweight"], f0)
                assert rel_err(alb, r[key + "/albedo"]) < 1e-10, key
                n += 1
            for hs in (0, 1):
                acc = 0
                for ig in range(wts.size):
                    f, _ = oracle.get_thermal_SH(nlevel, wno, nwno, ng, nt, r["sh/tlevel"], sl("dtau", ig), sl("tau", ig),
                                                 sl("w0", ig), sl("cosb", ig), sl("dtau_og", ig), sl("tau_og", ig),
                                                 sl("w0_og", ig), sl("w0_no_raman", ig), sl("cosb_og", ig), r["sh/plevel"],
                                                 geo["ubar1"], rs.copy(), stream, hs)
                    acc = acc + f * wts[ig]
                key = "sh/%s/thermal_s%d_hs%d" % (gname, stream, hs)
                assert rel_err(acc, r[key + "/flux_at_top"]) < 1e-10, key
                assert rel_err(oracle.compress_thermal(nwno, acc, geo["gweight"], geo["tweight"]), r[key + "/thermal"]) < 1e-10
    assert n == 12


@pytest.mark.parametrize("fam", ["r3d"])
def test_3d_inside_the_correlated_k_loop(fam, oracle):
    """justdoit.py:488-516 with the oracle in the solver's place, on the stored (nlayer|nlevel, nwno, 3, 3, 8) planes."""
    r, _, _ = _ck_rt()
    wts = r[fam + "/in/gauss_wts"]
    geo = {k: r["%s/geo/%s" % (fam, k)] for k in ("numg", "numt", "ubar0", "ubar1", "cos_theta", "gweight", "tweight")}
    ng, nt, ct = int(geo["numg"]), int(geo["numt"]), float(geo["cos_theta"])
    wno, rs, f0 = r[fam + "/in/wno"], r[fam + "/in/surf_reflect"], r[fam + "/in/F0PI"]
    nlevel, nwno = r[fam + "/in/tau"].shape[:2]
    pl = {k: r["%s/in/%s" % (fam, k)] for k in PLANES + ("w0_no_raman",)}
    tthg = (1.0, -1.0, 2.0, -0.5, 1.0)
    for sp, mp in ((3, 0), (0, 1), (1, 0)):
        acc = 0
        for ig in range(wts.size):
            x = oracle.get_reflected_3d(nlevel, wno, nwno, ng, nt, *[np.ascontiguousarray(pl[k][..., ig]) for k in PLANES], rs,
                                        geo["ubar0"], geo["ubar1"], ct, f0, sp, mp, *tthg)
            acc = acc + x * wts[ig]
        key = "%s/refl_sp%d_mp%d" % (fam, sp, mp)
        assert rel_err(acc, r[key + "/xint_at_top"]) < TOL, key
        assert rel_err(oracle.compress_disco(nwno, ct, acc, geo["gweight"], geo["tweight"], f0), r[key + "/albedo"]) < TOL
    for hs in (0, 1):
        acc = 0
        for ig in range(wts.size):
            f = oracle.get_thermal_3d(nlevel, wno, nwno, ng, nt, r[fam + "/in/tlevel"], np.ascontiguousarray(pl["dtau_og"][..., ig]),
                                      np.ascontiguousarray(pl["w0_no_raman"][..., ig]),
                                      np.ascontiguousarray(pl["cosb_og"][..., ig]), r[fam + "/in/plevel"], geo["ubar1"], rs, hs)
            acc = acc + f * wts[ig]
        key = "%s/therm_hs%d" % (fam, hs)
        assert rel_err(acc, r[key + "/flux_at_top"]) < TOL, key
        assert rel_err(oracle.compress_thermal(nwno, acc, geo["gweight"], geo["tweight"]), r[key + "/thermal"]) < TOL
