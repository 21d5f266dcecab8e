"""``spectrum()`` with correlated-k tables through the SH solvers and through the 3-D branch -- the combinations round 4
raised on ("not built").  The reference runs them as its plain loop (justdoit.py:256-307, 364-380, 488-516); expected
values: tests/golden/ck_rt.npz, outputs of the reference's own functions from table to spectrum (make_golden.py ck_rt:
get_pre_mix_ck + get_continuum + compute_opacity facet by facet, then the Gauss loop around the solvers)."""
import os
import sqlite3
import warnings

import numpy as np
import pytest

from helpers import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
DB = os.path.join(GOLDEN, "synthetic_opacities.db")
PAIRS = (("H2", "H2"), ("H2", "He"), ("H2", "CH4"))
TOL = 1e-9


@pytest.fixture(scope="module")
def fx():
    return (np.load(os.path.join(GOLDEN, "ck_rt.npz")), np.load(os.path.join(GOLDEN, "ck.npz")),
            np.load(os.path.join(GOLDEN, "optics.npz")))


def _db_tables():
    from picaso_amd import optics as px
    conn = sqlite3.connect(DB)
    ray = {m: px._convert_array(b) for m, b in conn.execute("SELECT molecule, opacity FROM rayleigh")}
    cont = {}
    for mol, t, blob in conn.execute("SELECT molecule, temperature, opacity FROM continuum"):
        cont.setdefault(mol, {})[float(t)] = px._convert_array(blob)
    conn.close()
    return ray, cont


def _opa(wno, gauss_wts, press, temps, nc_p, kappa, cia_temps, **extra):
    from picaso_amd import optics as px
    ray, cont = _db_tables()
    pressures = np.concatenate([press[:n] for n in nc_p])
    temps_flat = np.concatenate([[t] * n for t, n in zip(temps, nc_p)])
    return px.RetrieveCKs(wno, gauss_wts, pressures, temps_flat, nc_p, kappa,
                          continuum={a + b: cont[a + b] for a, b in PAIRS}, cia_temps=cia_temps, rayleigh_opa=ray, **extra)


def _case_1d(og, jdi, r, phase=0.0, ng=10, nt=1):
    case = jdi.inputs()
    case.phase_angle(phase, num_gangle=ng, num_tangle=nt)
    case.gravity(gravity=float(og["in/gravity"]))
    prof = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"]}
    for k in ("H2", "He", "H2O", "CH4"):
        prof[k] = og["in/mix/" + k]
    case.atmosphere(df=prof)
    case.clouds(df={"opd": og["in/cld_opd"], "w0": og["in/cld_w0"], "g0": og["in/cld_g0"]})
    case.surface_reflect(r["sh/surf_reflect"], og["in/wno"])
    return case


@pytest.mark.parametrize("stream", [2, 4])
@pytest.mark.parametrize("forms", [("TTHG", "TTHG", "TTHG", "on", "on", "on", "explicit"),
                                   ("OTHG", "OTHG", "OTHG", "on", "on", "on", "explicit"),
                                   ("isotropic", "TTHG", "OTHG", "on", "off", "on", "legendre")])
def test_sh_spectrum_on_correlated_k_tables(fx, stream, forms):
    """spectrum(rt_method='SH') on a RetrieveCKs object: the reference's loop over the Gauss points around
    get_reflected_SH / get_thermal_SH, from the k-table to albedo and thermal flux."""
    from picaso_amd import justdoit as jdi
    r, ck, og = fx
    opa = _opa(og["in/wno"], ck["in/gauss_wts"], ck["in/press"], ck["in/temps"], ck["in/nc_p"], ck["in/kappa"],
               ck["in/cia_temps"])
    opa.relative_flux = None
    case = _case_1d(og, jdi, r)
    opts = ["TTHG", "OTHG", "isotropic"]
    code = "s%d_f%d%d%d_r%d%d%d_sf%d" % (stream, opts.index(forms[0]), opts.index(forms[1]), opts.index(forms[2]),
                                        forms[3] == "on", forms[4] == "on", forms[5] == "on", forms[6] == "legendre")
    case.approx(raman="none", delta_eddington=True, rt_method="SH", stream=stream, w_single_form=forms[0],
                w_multi_form=forms[1], psingle_form=forms[2], w_single_rayleigh=forms[3], w_multi_rayleigh=forms[4],
                psingle_rayleigh=forms[5], single_form=forms[6])
    # F0PI of the fixture through the star slot (no star() call: relative flux handed over as data)
    case.inputs["star"].update(database="user", relative_flux=r["sh/F0PI"], radius=np.nan, semi_major=np.nan)
    out = case.spectrum(opa, calculation="reflected+thermal", full_output=True)
    key = "sh/g5/" + code
    assert rel_err(out["full_output"]["albedo_3d"], r[key + "/xint_at_top"]) < TOL
    assert rel_err(out["albedo"], r[key + "/albedo"]) < TOL
    assert case.inputs["hard_surface"] == 1            # surface_reflect() makes the surface hard, as in the reference
    assert rel_err(out["thermal"], r["sh/g5/thermal_s%d_hs1/thermal" % stream]) < TOL
    assert rel_err(out["full_output"]["thermal_3d"], r["sh/g5/thermal_s%d_hs1/flux_at_top" % stream]) < TOL
    case.inputs["hard_surface"] = 0
    soft = case.spectrum(opa, calculation="thermal")
    assert rel_err(soft["thermal"], r["sh/g5/thermal_s%d_hs0/thermal" % stream]) < TOL
    # one leg at a time: the same numbers (the two-stream overlap changes nothing).  Without full_output the default-options
    # launch leaves out the level planes (running products of the beam exponentials): within their rounding of the above
    both = case.spectrum(opa, calculation="reflected+thermal")
    alone = case.spectrum(opa, calculation="reflected")
    assert np.array_equal(alone["albedo"], both["albedo"])
    assert rel_err(alone["albedo"], out["albedo"]) < 1e-11 and rel_err(alone["albedo"], r[key + "/albedo"]) < TOL
    out = both
    # patchy clouds with SH: ignored, as in the reference (its blend exists for the Toon solver only) -- with a warning
    case.inputs["clouds"].update(do_holes=True, fhole=0.3, fthin_cld=0.1)
    with pytest.warns(UserWarning, match="do_holes has no effect"):
        holes = case.spectrum(opa, calculation="reflected")
    assert np.array_equal(holes["albedo"], out["albedo"])


def _case_3d(og, jdi, r, clouds="per_facet"):
    ng = nt = 3
    case = jdi.inputs()
    case.phase_angle(np.pi / 3, num_gangle=ng, num_tangle=nt)
    case.gravity(gravity=float(og["in/gravity"]))
    prof = {"pressure": og["in/plevel_bar"], "temperature": r["p3d/in/tlevel"]}
    for k in ("H2", "He", "H2O", "CH4"):
        prof[k] = og["in/mix/" + k]
    case.atmosphere_3d(prof)
    if clouds is not None:
        cf = r["p3d/in/cloud_scale"]
        cld = {k: np.repeat(og["in/cld_" + k][:, :, None, None], ng, 2).repeat(nt, 3) for k in ("opd", "w0", "g0")}
        cld["opd"] = cld["opd"] * cf[None, None]
        case.clouds_3d(cld)
    case.approx(raman="none", delta_eddington=True)
    case.surface_reflect(r["p3d/in/surf_reflect"], og["in/wno"])
    case.inputs["star"].update(database="user", relative_flux=r["p3d/in/F0PI"], radius=np.nan, semi_major=np.nan)
    return case


@pytest.mark.parametrize("phases", [(3, 0), (0, 1), (1, 0)])
def test_3d_spectrum_on_correlated_k_tables(fx, phases):
    """spectrum(dimension='3d') on an 8-Gauss-point premixed table, 3 x 3 facets with their own temperature columns and
    cloud optical depths: table -> facet-major planes -> the Gauss loop around get_reflected_3d / get_thermal_3d."""
    from picaso_amd import justdoit as jdi
    r, _, og = fx
    opa = _opa(og["in/wno"], r["p3d/in/gauss_wts"], r["p3d/in/press"], r["p3d/in/temps"], r["p3d/in/nc_p"],
               r["p3d/in/kappa"], r["p3d/in/cia_temps"])
    assert opa.ngauss == 8
    case = _case_3d(og, jdi, r)
    sp, mp = phases
    case.approx(raman="none", delta_eddington=True, single_phase=jdi.single_phase_options(False)[sp],
                multi_phase=jdi.multi_phase_options(False)[mp])
    out = case.spectrum(opa, calculation="reflected+thermal", dimension="3d", full_output=True)
    key = "p3d/refl_sp%d_mp%d" % (sp, mp)
    assert rel_err(out["full_output"]["albedo_3d"], r[key + "/xint_at_top"]) < TOL
    assert rel_err(out["albedo"], r[key + "/albedo"]) < TOL
    assert rel_err(out["full_output"]["thermal_3d"], r["p3d/therm_hs1/flux_at_top"]) < TOL       # surface_reflect(): hard
    assert rel_err(out["thermal"], r["p3d/therm_hs1/thermal"]) < TOL
    if phases == (3, 0):
        # all 13 planes written and read == the derived set (tau, tau_og, gcos2 re-derived in the kernels): the same bits
        full = case.spectrum(opa, calculation="reflected+thermal", dimension="3d", options=jdi.Options(all_planes=True))
        assert np.array_equal(full["albedo"], out["albedo"]) and np.array_equal(full["thermal"], out["thermal"])
        case.inputs["hard_surface"] = 0
        soft = case.spectrum(opa, calculation="thermal", dimension="3d")
        assert rel_err(soft["thermal"], r["p3d/therm_hs0/thermal"]) < TOL


def test_3d_ck_planes_of_one_facet(fx):
    """The facet-major opacity stage against the reference's per-facet compute_opacity(ngauss=8) (facet (2, 1))."""
    from picaso_amd import justdoit as jdi
    from picaso_amd import optics as px
    from picaso_amd.spectrum import _setup_atmosphere
    r, _, og = fx
    opa = _opa(og["in/wno"], r["p3d/in/gauss_wts"], r["p3d/in/press"], r["p3d/in/temps"], r["p3d/in/nc_p"],
               r["p3d/in/kappa"], r["p3d/in/cia_temps"])
    case = _case_3d(og, jdi, r)
    inp = case.inputs
    prof3 = inp["atmosphere"]["profile_3d"]
    nlv, nfac = len(prof3["pressure"]), 9
    prof_f = {k: (np.ascontiguousarray(np.broadcast_to(v.reshape(nlv, -1), (nlv, nfac))) if k == "temperature"
                  else v.reshape(nlv, -1)) for k, v in prof3.items()}
    atm_f = _setup_atmosphere(inp, opa, opa.wno, prof_f, None)
    pl = px.compute_opacity_facet_major_ck(atm_f, opa, 3, 3, stream=2, delta_eddington=True, raman=2,
                                           clouds_3d=inp["clouds"]["profile_3d"])
    f = 2 * 3 + 1
    for nm in ("dtau", "w0", "tau_og", "w0_no_raman", "cosb_og"):
        got = pl[nm].to_host()[f]
        assert got.shape == r["p3d/facet_2_1/" + nm].shape
        assert rel_err(got, r["p3d/facet_2_1/" + nm], 1e-300) < 1e-10, nm


def test_3d_ck_cloud_free_and_shared_cloud_equal_per_facet_planes(fx):
    """No cloud (two planes per leg, the rest re-derived) and one cloud table for the whole disk: the same spectrum as the
    same atmosphere handed over with explicit per-facet cloud arrays (zeros / tiled)."""
    from picaso_amd import justdoit as jdi
    r, _, og = fx
    opa = _opa(og["in/wno"], r["p3d/in/gauss_wts"], r["p3d/in/press"], r["p3d/in/temps"], r["p3d/in/nc_p"],
               r["p3d/in/kappa"], r["p3d/in/cia_temps"])
    ng = nt = 3
    clear = _case_3d(og, jdi, r, clouds=None).spectrum(opa, calculation="reflected+thermal", dimension="3d")
    zeros = _case_3d(og, jdi, r, clouds=None)
    zeros.clouds_3d({k: np.zeros(og["in/cld_opd"].shape + (ng, nt)) for k in ("opd", "w0", "g0")})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        z = zeros.spectrum(opa, calculation="reflected+thermal", dimension="3d")
    for k in ("albedo", "thermal"):
        assert np.isfinite(clear[k]).all() and rel_err(clear[k], z[k]) < 1e-12, k
    shared = _case_3d(og, jdi, r, clouds=None)
    shared.clouds_3d({k: og["in/cld_" + k] for k in ("opd", "w0", "g0")})
    tiled = _case_3d(og, jdi, r, clouds=None)
    tiled.clouds_3d({k: np.repeat(og["in/cld_" + k][:, :, None, None], ng, 2).repeat(nt, 3) for k in ("opd", "w0", "g0")})
    a = shared.spectrum(opa, calculation="reflected+thermal", dimension="3d")
    b = tiled.spectrum(opa, calculation="reflected+thermal", dimension="3d")
    for k in ("albedo", "thermal"):
        assert np.array_equal(a[k], b[k]), k


def test_3d_ck_on_the_fly_mixing_and_phase_curve(fx):
    """On-the-fly mixed k-tables in the 3-D branch (one resort-rebin launch over the tall atmosphere of all facets) equal
    the per-facet 1-D spectra of the same columns; phase_curve() on k-tables equals its phases run one by one."""
    from picaso_amd import justdoit as jdi
    r, ck, og = fx
    ng = nt = 2
    kap = {m: ck["fly/kappas/" + m] for m in ("H2O", "CH4", "H2")}
    opa = _opa(og["in/wno"], ck["in/gauss_wts"], ck["in/press"], ck["in/temps"], ck["in/nc_p"], None, ck["in/cia_temps"],
               kappas=kap, gauss_pts=ck["fly/gauss_pts"], on_fly=True)
    tfac = 1.0 + 0.1 * np.arange(ng * nt).reshape(ng, nt) / 4.0

    def case3(phase):
        c = jdi.inputs()
        c.phase_angle(phase, num_gangle=ng, num_tangle=nt)
        c.gravity(gravity=float(og["in/gravity"]))
        prof = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"][:, None, None] * tfac[None]}
        for k in ("H2", "He", "H2O", "CH4"):
            prof[k] = og["in/mix/" + k]
        c.atmosphere_3d(prof)
        c.approx(raman="none")
        return c
    out = case3(0.7).spectrum(opa, calculation="thermal", dimension="3d", full_output=True)
    f3 = out["full_output"]["thermal_3d"]
    assert np.isfinite(f3).all() and f3.min() > 0
    # facet (g, t) alone as a 1-D column under that facet's emission angle: get_thermal_3d differs from get_thermal_1d only
    # in its boundary factors (pi vs 2 pi), so compare through the solver-level 3-D call on 1-D planes instead:
    from picaso_amd import optics as px
    from picaso_amd import resident
    from picaso_amd.device import DeviceArray
    from picaso_amd.spectrum import _setup_atmosphere
    geom = case3(0.7).inputs["disco"]
    for g in range(ng):
        for t in range(nt):
            c1 = jdi.inputs()
            c1.phase_angle(0)
            c1.gravity(gravity=float(og["in/gravity"]))
            prof = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"] * tfac[g, t]}
            for k in ("H2", "He", "H2O", "CH4"):
                prof[k] = og["in/mix/" + k]
            c1.atmosphere(df=prof)
            c1.approx(raman="none")
            atm = _setup_atmosphere(c1.inputs, opa, opa.wno)
            opa.get_opacities(atm)
            pl = px.compute_opacity_resident(atm, opa, ngauss=opa.ngauss, stream=2, delta_eddington=True, test_mode=None,
                                             raman=2)
            nlevel, nwno = atm.c.nlevel, opa.nwno
            fx1 = DeviceArray((1, 1, nwno), opa.ctx)
            resident.thermal_3d_ck(opa.ctx, nlevel, DeviceArray.from_host(opa.wno, opa.ctx), nwno, opa.ngauss, 1, 1,
                                   atm.level["temperature"].reshape(nlevel, 1, 1), pl["dtau_og"], pl["w0_no_raman"],
                                   pl["cosb_og"], atm.level["pressure"].reshape(nlevel, 1, 1),
                                   np.array([[geom["ubar1"][g, t]]]), DeviceArray.zeros((nwno,), opa.ctx), 0, opa.gauss_wts,
                                   fx1)
            assert np.array_equal(fx1.to_host()[0, 0], f3[g, t]), (g, t)
    # phase curve on k-tables: every phase equals its own spectrum() call
    def profile(k):
        pr = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"][:, None, None] * (tfac[None] + 0.02 * k)}
        for m in ("H2", "He", "H2O", "CH4"):
            pr[m] = og["in/mix/" + m]
        return pr
    pc = jdi.inputs()
    pc.gravity(gravity=float(og["in/gravity"]))
    pc.approx(raman="none")
    phases = [0.0, 1.0, 2.5]
    pc.phase_curve_geometry("thermal", phases, num_gangle=ng, num_tangle=nt)
    pc.atmosphere_4d([profile(k) for k in range(len(phases))])
    res = pc.phase_curve(opa)
    for k, ph in enumerate(phases):
        one = jdi.inputs()
        one.gravity(gravity=float(og["in/gravity"]))
        one.approx(raman="none")
        one.phase_angle(0.0, num_gangle=ng, num_tangle=nt)          # thermal curves integrate over the phase-0 geometry
        one.atmosphere_3d(profile(k))
        want = one.spectrum(opa, calculation="thermal", dimension="3d")
        assert np.array_equal(res[ph]["thermal"], want["thermal"]), ph


@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
@pytest.mark.parametrize("calc", ["reflected", "thermal", "reflected+thermal"])
@pytest.mark.parametrize("cloud", [False, True])
def test_driver_runs_toon_spectra_on_k_tables(monkeypatch, fx, devices, calc, cloud):
    """A Toon spectrum on premixed k-tables through the C driver (picaso_spectrum_job.ngauss; round 5: it took the
    call-by-call path, a dozen ctypes calls for 0.18 ms of GPU work): picaso_opacity_gas_ck_dev ->
    picaso_compute_opacity_ck_dev -> picaso_get_reflected_1d_ck_dev || picaso_get_thermal_1d_ck_dev, whole grid and
    wavelength blocks, every output equal to spectrum.Spectrum's bit for bit."""
    from picaso_amd import justdoit as jdi
    r, ck, og = fx
    opa = _opa(og["in/wno"], ck["in/gauss_wts"], ck["in/press"], ck["in/temps"], ck["in/nc_p"], ck["in/kappa"],
               ck["in/cia_temps"])
    opa.relative_flux = None

    def make(k):
        case = jdi.inputs()
        case.phase_angle(0, num_gangle=5)
        case.gravity(gravity=float(og["in/gravity"]))
        prof = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"] * (1.0 + 0.01 * k)}
        for m in ("H2", "He", "H2O", "CH4"):
            prof[m] = og["in/mix/" + m]
        case.atmosphere(df=prof)
        if cloud:
            case.clouds(df={"opd": og["in/cld_opd"], "w0": og["in/cld_w0"], "g0": og["in/cld_g0"]})
        case.surface_reflect(r["sh/surf_reflect"], og["in/wno"])
        case.approx(raman="none", delta_eddington=True)
        return case
    monkeypatch.setenv("PICASO_AMD_NO_DRIVER", "1")
    want = [make(k).spectrum(opa, calculation=calc, devices=devices) for k in range(2)]
    assert "_driver_tables" not in opa.__dict__
    monkeypatch.delenv("PICASO_AMD_NO_DRIVER")
    got = [make(k).spectrum(opa, calculation=calc, devices=devices) for k in range(2)]
    assert len(opa.__dict__["_driver_tables"]) == 1
    for w, g in zip(want, got):
        assert set(w) == set(g)
        for key in w:
            if isinstance(w[key], np.ndarray):
                assert np.array_equal(w[key], g[key], equal_nan=True), key
            else:
                assert w[key] == g[key] or (w[key] != w[key] and g[key] != g[key]), key
        assert all(np.isfinite(v).all() for v in g.values() if isinstance(v, np.ndarray))
