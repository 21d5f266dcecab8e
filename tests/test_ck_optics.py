"""Premixed correlated-k opacity path and patchy clouds (rank 4).

(i) CPU: oracle/optics_oracle.py restatements of RetrieveCKs.get_pre_mix_ck / get_continuum and of
compute_opacity(ngauss=4) against tests/golden/ck.npz (outputs of the reference's own methods on a
synthetic ln(kappa) table, tests/golden/make_golden.py ck).  (ii) GPU: picaso_amd.optics.RetrieveCKs
+ compute_opacity against the same fixture, and inputs.spectrum() end to end against the
reference's Gauss-point / patchy-cloud loops restated with the CPU oracle."""
import os
import sqlite3

import numpy as np
import pytest

from helpers import GOLDEN, PLANES, rel_err
from test_optics import DB, NAMES, WEIGHTS, _close

CASES = ("de1_s2", "de0_s2", "de1_s4")
PAIRS = (("H2", "H2"), ("H2", "He"), ("H2", "CH4"))
TTHG = (1.0, -1.0, 2.0, -0.5, 1.0)


@pytest.fixture(scope="module")
def ck():
    return np.load(os.path.join(GOLDEN, "ck.npz"))


@pytest.fixture(scope="module")
def og():
    return np.load(os.path.join(GOLDEN, "optics.npz"))


def _db_tables():
    from picaso_amd import optics as px
    conn = sqlite3.connect(DB)
    ray = {m: px._convert_array(b) for m, b in conn.execute("SELECT molecule, opacity FROM rayleigh")}
    cont = {}
    for mol, t, blob in conn.execute("SELECT molecule, temperature, opacity FROM continuum"):
        cont.setdefault(mol, {})[float(t)] = px._convert_array(blob)
    conn.close()
    return ray, cont


def _layers(og):
    p = og["in/plevel_bar"] * 1e6
    t = og["in/tlevel"]
    mix = {k: 0.5 * (og["in/mix/" + k][1:] + og["in/mix/" + k][:-1]) for k in WEIGHTS}
    mmw_l = sum(og["in/mix/" + k] * WEIGHTS[k] for k in WEIGHTS)
    mmw = 0.5 * (mmw_l[1:] + mmw_l[:-1])
    g = float(og["in/gravity"])
    colden = og["in/colden"]                 # as the reference's ATMSETUP gives it (half-gravity end layers)
    tlayer = 0.5 * (t[1:] + t[:-1])
    player = np.sqrt(p[1:] * p[:-1]) / 1e6
    plev = p / 1e6
    A = (tlayer / (t[:-1] * t[1:])) * (t[1:] * plev[1:] - t[:-1] * plev[:-1]) / (plev[1:] - plev[:-1])
    B = (tlayer / (t[:-1] * t[1:])) * (t[:-1] - t[1:]) / (plev[1:] - plev[:-1])
    COEF1 = 8.31446261815324 * 273.15 ** 2 * .5E5 * (A * (plev[1:] ** 2 - plev[:-1] ** 2) + B * (2. / 3.) * (
        plev[1:] ** 3 - plev[:-1] ** 3)) / (1.01325 ** 2 * (g / 100.0) * tlayer * mmw)
    return dict(mix=mix, mmw=mmw, colden=colden, tlayer=tlayer, player=player, COEF1=COEF1)


def test_oracle_ck_against_reference(ck, og):
    from oracle import optics_oracle as oo
    L = _layers(og)
    ray, cont = _db_tables()
    mol = oo.pre_mix_ck(L["player"], L["tlayer"], ck["in/press"], ck["in/temps"], ck["in/nc_p"], ck["in/kappa"])
    assert _close(mol, ck["molecular_opa"], 1e-12)
    st = np.sort(ck["in/cia_temps"])
    cont_opa = {}
    for a, b in PAIRS:
        tab = np.stack([cont[a + b][t] for t in st])
        cont_opa[a + b] = oo.continuum_ck(L["tlayer"], st, tab)
        assert _close(cont_opa[a + b], ck["continuum_opa/" + a + b], 1e-12), a + b
    taugas = np.zeros(mol.shape)
    for a, b in PAIRS:                                           # optics.py:172-237
        taugas += (cont_opa[a + b] * (L["COEF1"] * L["mix"][a] * L["mix"][b])[:, None])[:, :, None]
    taugas += mol * (L["colden"] / L["mmw"])[:, None, None]       # optics.py:256-262
    tauray = np.zeros(mol.shape[:2])
    for m in ("H2", "He", "CH4", "H2O"):
        tauray += ray[m][None, :] * (L["colden"] * L["mix"][m] / L["mmw"])[:, None]
    b3 = lambda x: x[:, :, None]
    for key in CASES:
        de, s = bool(int(key[2])), int(key[-1])
        out = oo.compute_opacity(taugas, b3(tauray), b3(og["in/cld_opd"]), b3(og["in/cld_w0"]),
                                 b3(og["in/cld_g0"]), 0.99999, stream=s, delta_eddington=de)
        for nm, arr in zip(NAMES, out):
            ref = ck["%s/%s" % (key, nm)]
            assert _close(np.broadcast_to(arr, ref.shape), ref, 1e-11), (key, nm)


def _ck_class(ck, ctx=None, fly=False):
    from picaso_amd import optics as px
    ray, cont = _db_tables()
    wno = np.load(os.path.join(GOLDEN, "optics.npz"))["in/wno"]
    press, temps, nc_p = ck["in/press"], ck["in/temps"], ck["in/nc_p"]
    pressures = np.concatenate([press[:n] for n in nc_p])
    temps_flat = np.concatenate([[t] * n for t, n in zip(temps, nc_p)])
    extra = {}
    if fly:
        extra = dict(kappas={m: ck["fly/kappas/" + m] for m in ("H2O", "CH4", "H2")},
                     gauss_pts=ck["fly/gauss_pts"], on_fly=True)
    return px.RetrieveCKs(wno, ck["in/gauss_wts"], pressures, temps_flat, nc_p, ck["in/kappa"],
                          continuum={a + b: cont[a + b] for a, b in PAIRS}, cia_temps=ck["in/cia_temps"],
                          rayleigh_opa=ray, **extra)


def _case(og, jdi, de=True, order=("H2", "He", "H2O", "CH4")):
    case = jdi.inputs()
    case.phase_angle(0)
    case.gravity(gravity=float(og["in/gravity"]))
    prof = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"]}
    for k in order:
        prof[k] = og["in/mix/" + k]
    case.atmosphere(df=prof)
    case.clouds(df={"opd": og["in/cld_opd"], "w0": og["in/cld_w0"], "g0": og["in/cld_g0"]})
    case.approx(raman="none", delta_eddington=de)
    return case


@pytest.mark.gpu
def test_gpu_ck_opacities_and_mixing(ck, og):
    from picaso_amd import justdoit as jdi
    from picaso_amd import optics as px
    from picaso_amd.atmsetup import ATMSETUP
    opa = _ck_class(ck)
    assert opa.ngauss == 4
    for key in CASES:
        de, s = bool(int(key[2])), int(key[-1])
        case = _case(og, jdi, de)
        atm = ATMSETUP(case.inputs)
        atm.planet.gravity = case.inputs["planet"]["gravity"]
        atm.get_profile(); atm.get_mmw(); atm.get_altitude(); atm.get_column_density()
        atm.get_needed_continuum(opa.rayleigh_molecules, opa.avail_continuum)
        atm.get_clouds(opa.wno)
        opa.get_opacities(atm)
        if key == CASES[0]:
            assert _close(opa.get_molecular_opa(), ck["molecular_opa"], 1e-11)
            for a, b in PAIRS:
                assert _close(opa.continuum_opa[a + b], ck["continuum_opa/" + a + b], 1e-11), a + b
        out = px.compute_opacity(atm, opa, ngauss=4, stream=s, delta_eddington=de, raman=2, test_mode=None)
        assert len(out) == 13
        for nm, arr in zip(NAMES, out):
            assert arr.shape == ck["%s/%s" % (key, nm)].shape
            assert _close(arr, ck["%s/%s" % (key, nm)], 1e-10), (key, nm)
    with pytest.raises(Exception, match="Gauss points"):
        px.compute_opacity(atm, opa, ngauss=1, test_mode=None)


def _oracle_loop(oracle, planes, wts, nlevel, wno, nwno, u0, u1, rs, tlevel, plevel):
    """justdoit.py:256-307 / 328-380 with the CPU oracle on the slices plane[:, :, ig]."""
    x = f = 0.0
    for ig in range(len(wts)):
        P = {k: np.ascontiguousarray(planes[k][:, :, ig]) for k in planes}
        xi, _ = oracle.get_reflected_1d(nlevel, wno, nwno, 5, 1, *[P[k] for k in PLANES], rs, u0, u1, 1.0,
                                        np.ones(nwno), 3, 0, *TTHG)
        fi, _ = oracle.get_thermal_1d(nlevel, wno, nwno, 5, 1, tlevel, P["dtau_og"], P["w0_no_raman"],
                                      P["cosb_og"], plevel, u1, np.full(nwno, rs), 1, wno * 0, 0)
        x, f = x + xi * wts[ig], f + fi * wts[ig]
    return x, f


@pytest.mark.gpu
def test_gpu_ck_spectrum_end_to_end(ck, og, oracle):
    """inputs.spectrum() with a 4-point correlated-k table: one batched launch over nwno*ngauss
    columns per solver, against the reference's Gauss-point loop restated with the CPU oracle on
    the reference's own compute_opacity(ngauss=4) planes."""
    from picaso_amd import disco
    from picaso_amd import justdoit as jdi
    opa = _ck_class(ck)
    case = _case(og, jdi, True)
    case.surface_reflect(0.2)
    out = case.spectrum(opa, calculation="reflected+thermal", full_output=True)
    planes = {nm: ck["de1_s2/" + nm] for nm in NAMES}
    nlevel, nwno = planes["tau"].shape[:2]
    g, gw, t, tw = disco.get_angles_1d(5)
    u0, u1, ct, _, _ = disco.compute_disco(5, 1, g, t, 0.0)
    x, f = _oracle_loop(oracle, planes, ck["in/gauss_wts"], nlevel, opa.wno, nwno, u0, u1, 0.2,
                        og["in/tlevel"], og["in/plevel_bar"] * 1e6)
    assert rel_err(out["full_output"]["albedo_3d"], x) < 1e-8
    assert rel_err(out["albedo"], oracle.compress_disco(nwno, 1.0, x, gw, tw, np.ones(nwno))) < 1e-8
    assert rel_err(out["thermal"], oracle.compress_thermal(nwno, f, gw, tw)) < 1e-8


@pytest.mark.gpu
def test_gpu_patchy_clouds_end_to_end(og, oracle):
    """clouds(do_holes=True, fhole, fthin_cld): cloudy and thinned columns blended as
    (1-fhole)*cloudy + fhole*clear (justdoit.py:248-252, 287-305, 346-361)."""
    from oracle import optics_oracle as oo
    from picaso_amd import disco
    from picaso_amd import justdoit as jdi
    fhole, fthin = 0.3, 0.1
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    case = _case(og, jdi, True)
    case.clouds(df={"opd": og["in/cld_opd"], "w0": og["in/cld_w0"], "g0": og["in/cld_g0"]}, do_holes=True,
                fhole=fhole, fthin_cld=fthin)
    case.surface_reflect(0.2)
    out = case.spectrum(opa, calculation="reflected+thermal", full_output=True)
    # reference planes of the cloudy column from the fixture; thinned column from the oracle mixing
    # with the reference's TAUGAS/TAURAY (recovered from the fixture planes of the un-thinned case)
    key = "linear/de0_s2_r2_tmnone"
    dtau, taucld = og[key + "/dtau_og"], og["in/cld_opd"]
    fray = og[key + "/ftau_ray"]
    with np.errstate(invalid="ignore", divide="ignore"):
        tauray = np.where(fray > 0, fray * og["in/cld_w0"] * taucld / np.where(fray < 1, 1 - fray, 1.0), 0.0)
    # cloud-free layers have ftau_ray = 1: take Rayleigh from w0_no_raman there instead
    tauray = np.where(taucld > 0, tauray, og[key + "/w0_no_raman"] * dtau / 0.99999)
    taugas = dtau - tauray - taucld
    nlevel, nwno = og[key + "/tau"].shape
    g, gw, t, tw = disco.get_angles_1d(5)
    u0, u1, ct, _, _ = disco.compute_disco(5, 1, g, t, 0.0)
    xs, fs = [], []
    for thin in (1.0, fthin):
        P = dict(zip(NAMES, oo.compute_opacity(taugas, tauray, thin * taucld, og["in/cld_w0"],
                                               og["in/cld_g0"], 0.99999, stream=2, delta_eddington=True)))
        xi, _ = oracle.get_reflected_1d(nlevel, opa.wno, nwno, 5, 1, *[P[k] for k in PLANES], 0.2, u0, u1,
                                        1.0, np.ones(nwno), 3, 0, *TTHG)
        fi, _ = oracle.get_thermal_1d(nlevel, opa.wno, nwno, 5, 1, og["in/tlevel"], P["dtau_og"],
                                      P["w0_no_raman"], P["cosb_og"], og["in/plevel_bar"] * 1e6, u1,
                                      np.full(nwno, 0.2), 1, opa.wno * 0, 0)
        xs.append(xi)
        fs.append(fi)
    x = (1.0 - fhole) * xs[0] + fhole * xs[1]
    f = (1.0 - fhole) * fs[0] + fhole * fs[1]
    assert rel_err(out["full_output"]["albedo_3d"], x) < 1e-7
    assert rel_err(out["albedo"], oracle.compress_disco(nwno, 1.0, x, gw, tw, np.ones(nwno))) < 1e-7
    assert rel_err(out["thermal"], oracle.compress_thermal(nwno, f, gw, tw)) < 1e-7


@pytest.mark.gpu
def test_gpu_3d_spectrum_end_to_end(og, oracle):
    """spectrum(dimension='3d'): per-facet opacities -> facet-strided planes on the device ->
    get_reflected_3d / get_thermal_3d -> disk integration, against the oracle chain.  Facets share
    the gas column and differ in cloud optical depth (x facet factor)."""
    from oracle import optics_oracle as oo
    from picaso_amd import disco
    from picaso_amd import justdoit as jdi
    ng, nt = 2, 3
    fac = 0.25 + np.arange(ng * nt).reshape(ng, nt) / 4.0
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    case = jdi.inputs()
    case.phase_angle(np.pi / 3, num_gangle=ng, num_tangle=nt)
    case.gravity(gravity=float(og["in/gravity"]))
    nlevel = len(og["in/tlevel"])
    prof = {"pressure": og["in/plevel_bar"], "temperature": np.repeat(og["in/tlevel"][:, None, None], ng, 1).repeat(nt, 2)}
    for k in ("H2", "He", "H2O", "CH4"):
        prof[k] = og["in/mix/" + k]
    case.atmosphere_3d(prof)
    cld = {k: np.repeat(og["in/cld_" + k][:, :, None, None], ng, 2).repeat(nt, 3) for k in ("opd", "w0", "g0")}
    cld["opd"] = cld["opd"] * fac[None, None]
    case.clouds_3d(cld)
    case.approx(raman="none", delta_eddington=True)
    case.surface_reflect(0.2)
    out = case.spectrum(opa, calculation="reflected+thermal", dimension="3d", full_output=True)
    # oracle chain
    key = "linear/de0_s2_r2_tmnone"
    dtau, taucld = og[key + "/dtau_og"], og["in/cld_opd"]
    fray = og[key + "/ftau_ray"]
    with np.errstate(invalid="ignore", divide="ignore"):
        tauray = np.where(fray > 0, fray * og["in/cld_w0"] * taucld / np.where(fray < 1, 1 - fray, 1.0), 0.0)
    tauray = np.where(taucld > 0, tauray, og[key + "/w0_no_raman"] * dtau / 0.99999)
    taugas = dtau - tauray - taucld
    nwno = dtau.shape[1]
    P3 = {nm: [] for nm in NAMES}
    for g in range(ng):
        for t in range(nt):
            P = dict(zip(NAMES, oo.compute_opacity(taugas, tauray, fac[g, t] * taucld, og["in/cld_w0"],
                                                   og["in/cld_g0"], 0.99999, stream=2, delta_eddington=True)))
            for nm in NAMES:
                P3[nm].append(P[nm])
    P3 = {nm: np.ascontiguousarray(np.stack(v, axis=2).reshape(v[0].shape + (ng, nt))) for nm, v in P3.items()}
    gg, gw, tt, tw = disco.get_angles_3d(ng, nt)
    u0, u1, ct, _, _ = disco.compute_disco(ng, nt, gg, tt, np.pi / 3)
    x = oracle.get_reflected_3d(nlevel, opa.wno, nwno, ng, nt, *[P3[k] for k in PLANES], 0.2, u0, u1, ct,
                                np.ones(nwno), 3, 0, *TTHG)
    assert rel_err(out["full_output"]["albedo_3d"], x) < 1e-7
    assert rel_err(out["albedo"], oracle.compress_disco(nwno, ct, x, gw, tw, np.ones(nwno))) < 1e-7
    tl3 = np.repeat(og["in/tlevel"][:, None, None], ng, 1).repeat(nt, 2)
    pl3 = np.repeat((og["in/plevel_bar"] * 1e6)[:, None, None], ng, 1).repeat(nt, 2)
    f = oracle.get_thermal_3d(nlevel, opa.wno, nwno, ng, nt, tl3, P3["dtau_og"], P3["w0_no_raman"],
                              P3["cosb_og"], pl3, u1, np.full(nwno, 0.2), 1)
    assert rel_err(out["full_output"]["thermal_3d"], f) < 1e-7
    assert rel_err(out["thermal"], oracle.compress_thermal(nwno, f, gw, tw)) < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("raman,radius", [("none", False), ("oklopcic", False), ("none", True), ("oklopcic", True)])
def test_gpu_3d_batched_facets_equal_per_facet_loop(og, raman, radius, monkeypatch):
    """The 3-D path sets all facets up in ONE facet-form ATMSETUP and gathers their gas optical depths in
    one batched launch; the per-facet loop (one ATMSETUP, bracket search and launch per facet, the
    reference's structure justdoit.py:437-471) gives bit-identical spectra -- facets with their own
    temperature and water profiles, constant and altitude-dependent gravity, with and without the
    per-facet Raman plane."""
    from picaso_amd import justdoit as jdi
    ng, nt = 4, 3
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    if raman == "oklopcic":
        gold = np.load(os.path.join(GOLDEN, "optics.npz"))
        opa.raman_stellar_shifts = gold["in/raman_shifts"]
        opa.raman_db = {"c": gold["in/raman_c"], "ji": gold["in/raman_ji"], "deltanu": gold["in/raman_deltanu"]}
    pert = 1.0 + 0.15 * np.cos(np.arange(ng * nt).reshape(ng, nt))

    def run():
        case = jdi.inputs()
        case.phase_angle(np.pi / 4, num_gangle=ng, num_tangle=nt)
        if radius:
            case.gravity(radius=7.0e9, mass=1.9e30)
        else:
            case.gravity(gravity=float(og["in/gravity"]))
        prof = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"][:, None, None] * pert[None]}
        for k in ("H2", "He", "CH4"):
            prof[k] = og["in/mix/" + k]
        prof["H2O"] = og["in/mix/H2O"][:, None, None] * (2.0 - pert)[None]
        case.atmosphere_3d(prof)
        case.approx(raman=raman, delta_eddington=True)
        case.surface_reflect(0.1)
        return case.spectrum(opa, calculation="reflected+thermal", dimension="3d", full_output=True)
    a = run()
    monkeypatch.setenv("PICASO_AMD_FACET_LOOP", "1")
    b = run()
    for k in ("albedo", "thermal"):
        assert np.all(np.isfinite(a[k]))
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["full_output"]["albedo_3d"], b["full_output"]["albedo_3d"])
    assert np.array_equal(a["full_output"]["thermal_3d"], b["full_output"]["thermal_3d"])


@pytest.mark.gpu
@pytest.mark.parametrize("raman", ["pollack", "oklopcic"])
@pytest.mark.parametrize("dimension", ["1d", "3d"])
def test_gpu_raman_on_the_device_equals_host_planes(og, dimension, raman, monkeypatch, tmp_path):
    """raman='pollack' (the reference's default, justdoit.py:4636): the factor depends on the wavelength only. The
    reference tiles the table over the layers (optics.py:296-298, np.repeat) and, in 3-D, does so once per facet;
    here ONE row of nwno values stays on the opacity object and the mixing kernels read it for every layer and facet
    (`raman_rows = 0`).  raman='oklopcic': the factor plane of compute_raman (optics.py:434-494) is formed by
    `picaso_raman_oklopcic_dev` from resident per-transition tables instead of ~1 s of numpy per call.  Both
    bit-identical to the host planes (PICASO_AMD_RAMAN_PLANES=1), also through the per-facet loop and in wavelength
    blocks."""
    from picaso_amd import justdoit as jdi
    ng, nt = 3, 2
    g = np.load(os.path.join(GOLDEN, "raman_pollack.npz"))
    (tmp_path / "opacities").mkdir()
    np.savetxt(tmp_path / "opacities" / "raman_fortran.txt", np.column_stack([g["table/w"], g["table/f"]]), fmt="%.17g")
    monkeypatch.setenv("picaso_refdata", str(tmp_path))
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    if raman == "oklopcic":
        opa.raman_stellar_shifts = og["in/raman_shifts"]
        opa.raman_db = {"c": og["in/raman_c"], "ji": og["in/raman_ji"], "deltanu": og["in/raman_deltanu"]}
    pert = 1.0 + 0.1 * np.cos(np.arange(ng * nt).reshape(ng, nt))

    def run(devices=None):
        case = jdi.inputs()                                    # no approx(): Raman is Pollack's table by default
        if raman != "pollack":
            case.approx(raman=raman)
        case.gravity(gravity=float(og["in/gravity"]))
        case.surface_reflect(0.1)
        if dimension == "1d":
            case.phase_angle(0)
            prof = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"]}
            for k in ("H2", "He", "CH4", "H2O"):
                prof[k] = og["in/mix/" + k]
            case.atmosphere(df=prof)
            case.clouds(df={"opd": og["in/cld_opd"], "w0": og["in/cld_w0"], "g0": og["in/cld_g0"]})
        else:
            case.phase_angle(np.pi / 3, num_gangle=ng, num_tangle=nt)
            prof = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"][:, None, None] * pert[None]}
            for k in ("H2", "He", "CH4", "H2O"):
                prof[k] = og["in/mix/" + k]
            case.atmosphere_3d(prof)
        return case.spectrum(opa, calculation="reflected+thermal", dimension=dimension, devices=devices)
    row = run()
    blocks = run(devices=[0, 0])
    monkeypatch.setenv("PICASO_AMD_RAMAN_PLANES", "1")
    planes = run()
    if dimension == "3d":
        monkeypatch.setenv("PICASO_AMD_FACET_LOOP", "1")
        loop = run()
    monkeypatch.delenv("PICASO_AMD_RAMAN_PLANES")
    case = jdi.inputs()
    for k in ("albedo", "thermal"):
        assert np.isfinite(row[k]).all()
        assert np.array_equal(row[k], planes[k]) and np.array_equal(row[k], blocks[k]), k
        if dimension == "3d":
            assert np.array_equal(row[k], loop[k]), k
    # and it is not the Raman-less spectrum
    monkeypatch.delenv("PICASO_AMD_FACET_LOOP", raising=False)
    if dimension == "1d":
        case.phase_angle(0)
        case.gravity(gravity=float(og["in/gravity"]))
        case.surface_reflect(0.1)
        prof = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"]}
        for k in ("H2", "He", "CH4", "H2O"):
            prof[k] = og["in/mix/" + k]
        case.atmosphere(df=prof)
        case.clouds(df={"opd": og["in/cld_opd"], "w0": og["in/cld_w0"], "g0": og["in/cld_g0"]})
        case.approx(raman="none")
        assert not np.array_equal(case.spectrum(opa, calculation="reflected")["albedo"], row["albedo"])


@pytest.mark.gpu
@pytest.mark.parametrize("cloudy,raman,delta", [(False, "none", True), (False, "oklopcic", True), (True, "none", True),
                                                (True, "oklopcic", False), (False, "none", False)])
def test_gpu_3d_planes_rederived_in_the_solvers_bit_identical(og, cloudy, raman, delta, monkeypatch):
    """The 3-D path does not write the planes the solvers can re-derive exactly (level optical depths, gcos2;
    without cloud also cosb, ftau_cld, ftau_ray, the *_og twins and -- raman='none' -- w0_no_raman): the
    spectra are bit-identical to those from the full set of 13 planes (PICASO_AMD_ALL_PLANES=1)."""
    from picaso_amd import justdoit as jdi
    ng, nt = 3, 2
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    if raman == "oklopcic":
        gold = np.load(os.path.join(GOLDEN, "optics.npz"))
        opa.raman_stellar_shifts = gold["in/raman_shifts"]
        opa.raman_db = {"c": gold["in/raman_c"], "ji": gold["in/raman_ji"], "deltanu": gold["in/raman_deltanu"]}
    pert = 1.0 + 0.1 * np.cos(np.arange(ng * nt).reshape(ng, nt))

    def run(calc):
        case = jdi.inputs()
        case.phase_angle(np.pi / 3, num_gangle=ng, num_tangle=nt)
        case.gravity(gravity=float(og["in/gravity"]))
        prof = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"][:, None, None] * pert[None]}
        for k in ("H2", "He", "H2O", "CH4"):
            prof[k] = og["in/mix/" + k]
        case.atmosphere_3d(prof)
        if cloudy:
            cld = {k: np.repeat(og["in/cld_" + k][:, :, None, None], ng, 2).repeat(nt, 3) for k in ("opd", "w0", "g0")}
            cld["opd"] = cld["opd"] * pert[None, None]
            case.clouds_3d(cld)
        case.approx(raman=raman, delta_eddington=delta)
        case.surface_reflect(0.15)
        return case.spectrum(opa, calculation=calc, dimension="3d", full_output=True)
    for calc in ("reflected+thermal", "thermal", "reflected"):
        monkeypatch.delenv("PICASO_AMD_ALL_PLANES", raising=False)
        monkeypatch.delenv("PICASO_AMD_MIX_DIRECT", raising=False)
        a = run(calc)
        monkeypatch.setenv("PICASO_AMD_ALL_PLANES", "1")
        monkeypatch.setenv("PICASO_AMD_MIX_DIRECT", "1")      # and the mixing kernel without the LDS-staged transposition
        b = run(calc)
        for k, k3 in (("albedo", "albedo_3d"), ("thermal", "thermal_3d")):
            if k in a:
                assert np.all(np.isfinite(a[k])) and np.any(a[k] > 0)
                assert np.array_equal(a[k], b[k]), (calc, k)
                assert np.array_equal(a["full_output"][k3], b["full_output"][k3]), (calc, k3)


# ---- on-the-fly gas mixing through the class (reference optics.py:1164-1278) ----------------------
def _fly_atm(og):
    import types
    L = _layers(og)
    atm = types.SimpleNamespace()
    atm.c = types.SimpleNamespace(nlayer=len(L["tlayer"]), pconv=1e6)
    atm.layer = {"temperature": L["tlayer"], "pressure": L["player"] * 1e6, "mixingratios": L["mix"]}
    atm.molecules = np.array(["H2O", "CH4", "H2"])
    return atm


def test_mixing_indices_against_reference(ck, og):
    """get_mixing_indices (host search) on a bare object: indices and weights of the reference."""
    import types
    from picaso_amd import optics as px
    press, temps, nc_p = ck["in/press"], ck["in/temps"], ck["in/nc_p"]
    bare = types.SimpleNamespace(pressures=np.unique(np.concatenate([press[:n] for n in nc_p])),
                                 temps=np.unique(temps), nc_p=nc_p)
    idx, t_i, p_i = px.RetrieveCKs.get_mixing_indices(bare, _fly_atm(og))
    assert np.array_equal(idx, ck["fly/indices"])
    assert _close(t_i, ck["fly/t_interp"], 1e-14) and _close(p_i, ck["fly/p_interp"], 1e-14)


def test_oracle_gasesfly_molecular_opa(ck, og, oracle):
    """Oracle mixing + the reference's bilinear step (optics.py:1191-1197) = its molecular_opa."""
    atm = _fly_atm(og)
    kappas = [ck["fly/kappas/" + m] for m in atm.molecules]
    mixes = [atm.layer["mixingratios"][m] for m in atm.molecules]
    xg, wg = ck["fly/gauss_pts"], ck["in/gauss_wts"]
    km = oracle.mix_all_gases_gasesfly(kappas, mixes, xg, wg, ck["fly/indices"])
    t, p = ck["fly/t_interp"][:, None, None], ck["fly/p_interp"][:, None, None]
    kap = (1 - t) * (1 - p) * km[..., 0] + t * (1 - p) * km[..., 1] + t * p * km[..., 3] + (1 - t) * p * km[..., 2]
    assert _close(np.exp(kap) * 6.02214086e+23, ck["fly/molecular_opa"], 1e-11)


@pytest.mark.gpu
def test_gpu_gasesfly_opacities_and_mixing(ck, og):
    """RetrieveCKs with per-gas tables: k_ckmix + ln-bilinear interpolation against the reference's
    mix_my_opacities_gasesfly (incl. exclude_mol) and compute_opacity on the mixed table."""
    from picaso_amd import justdoit as jdi
    from picaso_amd import optics as px
    opa = _ck_class(ck, fly=True)
    assert list(opa.molecules) == ["H2O", "CH4", "H2"]
    # gases are mixed in the order of the profile columns (atmosphere.molecules); the fixture's is
    # H2O, CH4, H2 and sequential resort-rebin is not commutative
    case = _case(og, jdi, True, order=("H2O", "CH4", "H2", "He"))
    atm = jdi._setup_atmosphere(case.inputs, opa, opa.wno)
    assert list(atm.molecules) == ["H2O", "CH4", "H2"]
    opa.get_opacities(atm)                                    # bound to get_opacities_deq_onfly
    assert _close(opa.get_molecular_opa(), ck["fly/molecular_opa"], 1e-10)
    for pr in ("H2H2", "H2He", "H2CH4"):
        assert _close(opa.continuum_opa[pr], ck["continuum_opa/" + pr], 1e-12), pr
    out = px.compute_opacity(atm, opa, ngauss=4, stream=2, delta_eddington=True, raman=2, test_mode=None)
    for nm, arr in zip(NAMES, out):
        ref = ck["fly/de1_s2/" + nm]
        assert _close(np.broadcast_to(arr, ref.shape), ref, 1e-9), nm
    opa.get_opacities(atm, exclude_mol={"H2O": 1, "CH4": 0, "H2": 1})
    assert _close(opa.get_molecular_opa(), ck["fly/molecular_opa_noCH4"], 1e-10)
    # the premixed table is still reachable
    opa.get_opacities_preweighted(atm)
    assert _close(opa.get_molecular_opa(), ck["molecular_opa"], 1e-10)


@pytest.mark.gpu
def test_gpu_gasesfly_spectrum_runs_through_picaso(ck, og):
    from picaso_amd import justdoit as jdi
    opa = _ck_class(ck, fly=True)
    out = _case(og, jdi, True).spectrum(opa, calculation="reflected+thermal")
    pre = _case(og, jdi, True).spectrum(_ck_class(ck), calculation="reflected+thermal")
    assert np.isfinite(out["albedo"]).all() and np.isfinite(out["thermal"]).all()
    assert not np.allclose(out["albedo"], pre["albedo"])      # different (synthetic) gas tables


@pytest.mark.gpu
@pytest.mark.parametrize("clouds", [None, "shared", "per_facet"])
@pytest.mark.parametrize("calc", ["reflected", "thermal"])
def test_gpu_phase_curve_equals_single_phase_runs(og, calc, clouds):
    """phase_curve(): every phase of phase_curve_geometry with its own facet profiles, against the
    same phases run one at a time through phase_angle() + atmosphere_3d() + spectrum(dimension='3d')
    (thermal phase curves integrate over the phase-0 geometry, justdoit.py:1648-1653).  ``clouds``: a cloud map per
    phase (``clouds_by_phase``) as tables on a 7-point wavenumber grid of their own, one for the disk or one per facet
    -- interpolated inside the fused opacity launch of each phase, solved in the batched launch of the chunk."""
    from picaso_amd import justdoit as jdi
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    ng, nt = 3, 2
    phases = [0.0, 0.9, 2.0]
    nlevel = len(og["in/tlevel"])
    rng = np.random.default_rng(8)
    wn = np.linspace(opa.wno[0] * 0.98, opa.wno[-1] * 1.01, 7)

    def cloud_map(k):
        if clouds is None:
            return None
        shape = (nlevel - 1, 7) if clouds == "shared" else (nlevel - 1, 7, ng, nt)
        c = {"opd": (0.1 + 0.05 * k) * rng.random(shape), "w0": 0.6 + 0.3 * rng.random(shape), "g0": 0.7 * rng.random(shape),
             "wavenumber": wn}
        c["opd"][:4] = 0.0
        return c
    maps = [cloud_map(k) for k in range(len(phases))]

    def profile(k):
        dT = 40.0 * k * np.cos(np.arange(ng))[None, :, None] * np.ones((1, 1, nt))
        pr = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"][:, None, None] + dT}
        for m in ("H2", "He", "H2O", "CH4"):
            pr[m] = og["in/mix/" + m]
        return pr

    case = jdi.inputs()
    case.gravity(gravity=float(og["in/gravity"]))
    case.approx(raman="none")
    case.phase_curve_geometry(calc, phases, num_gangle=ng, num_tangle=nt)
    case.atmosphere_4d([profile(k) for k in range(len(phases))])
    curve = case.phase_curve(opa, clouds_by_phase=maps if clouds else None)
    assert list(curve.keys()) == phases and case.inputs["phase_angle"] == phases
    key = "albedo" if calc == "reflected" else "thermal"
    for k, ph in enumerate(phases):
        one = jdi.inputs()
        one.gravity(gravity=float(og["in/gravity"]))
        one.approx(raman="none")
        one.phase_angle(ph if calc == "reflected" else 0.0, num_gangle=ng, num_tangle=nt)
        one.atmosphere_3d(profile(k))
        if clouds:
            one.clouds_3d({a: (b.copy() if a != "wavenumber" else b) for a, b in maps[k].items() if not a.startswith("_")})
        want = one.spectrum(opa, calculation=calc, dimension="3d")
        assert np.array_equal(curve[ph][key], want[key]), (calc, ph)
    assert not np.array_equal(curve[phases[0]][key], curve[phases[2]][key])


@pytest.mark.gpu
def test_gpu_phase_curve_against_oracle_solver(og, oracle):
    """phase_curve() with every phase enqueued before the first copy back: per phase, the planes of the
    product's own opacity stage (facet-form ATMSETUP + batched gas stage, pinned to the reference by
    the tests above) go through the CPU oracle's get_reflected_3d / compress_disco (and get_thermal_3d /
    compress_thermal for a thermal curve) -- an independent check of the solver and disk integration of
    every phase, next to the self-consistency test above."""
    from picaso_amd import disco, optics
    from picaso_amd import justdoit as jdi
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    ng, nt = 4, 3
    phases = [0.0, 0.7, 1.9, 3.0, 4.4]
    nlevel = len(og["in/tlevel"])

    def profile(k):
        dT = 30.0 * (k + 1) * np.cos(np.arange(ng * nt).reshape(ng, nt) + k)[None]
        pr = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"][:, None, None] + dT}
        for m in ("H2", "He", "H2O", "CH4"):
            pr[m] = og["in/mix/" + m]
        return pr
    for calc in ("reflected", "thermal"):
        case = jdi.inputs()
        case.gravity(gravity=float(og["in/gravity"]))
        case.approx(raman="none")
        case.surface_reflect(0.15)
        case.phase_curve_geometry(calc, phases, num_gangle=ng, num_tangle=nt)
        case.atmosphere_4d([profile(k) for k in range(len(phases))])
        curve = case.phase_curve(opa, full_output=True)
        gg, gw, tt, tw = disco.get_angles_3d(ng, nt)
        for k, ph in enumerate(phases):
            prof3 = {kk: np.asarray(v, dtype=float) for kk, v in profile(k).items()}
            prof_f = {kk: (np.ascontiguousarray(np.broadcast_to(v.reshape(nlevel, -1), (nlevel, ng * nt)))
                           if kk == "temperature" else v.reshape(nlevel, -1)) for kk, v in prof3.items()}
            atm_f = jdi._setup_atmosphere(case.inputs, opa, opa.wno, prof_f, None)
            P = optics.compute_opacity_facets(atm_f, opa, ng, nt, stream=2, delta_eddington=True, test_mode=None,
                                              raman=2, clouds_3d=None, exclude_mol=1)
            P = {kk: v.to_host() for kk, v in P.items()}
            nwno = opa.nwno
            u0, u1, ct, _, _ = disco.compute_disco(ng, nt, gg, tt, ph if calc == "reflected" else 0.0)
            if calc == "reflected":
                x = oracle.get_reflected_3d(nlevel, opa.wno, nwno, ng, nt, *[P[kk] for kk in PLANES], 0.15, u0, u1,
                                            ct, np.ones(nwno), 3, 0, *TTHG)
                assert rel_err(curve[ph]["full_output"]["albedo_3d"], x) < 1e-8, ph
                assert rel_err(curve[ph]["albedo"], oracle.compress_disco(nwno, ct, x, gw, tw, np.ones(nwno))) < 1e-8
            else:
                tl3 = prof3["temperature"]
                pl3 = np.repeat((og["in/plevel_bar"] * 1e6)[:, None, None], ng, 1).repeat(nt, 2)
                f = oracle.get_thermal_3d(nlevel, opa.wno, nwno, ng, nt, tl3, P["dtau_og"], P["w0_no_raman"],
                                          P["cosb_og"], pl3, u1, np.full(nwno, 0.15), 1)
                assert rel_err(curve[ph]["full_output"]["thermal_3d"], f) < 1e-8, ph
                assert rel_err(curve[ph]["thermal"], oracle.compress_thermal(nwno, f, gw, tw)) < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("nlayer", [1, 29, 30, 31, 90])
def test_gpu_raman_oklopcic_plane_is_compute_raman(og, nlayer):
    """picaso_raman_oklopcic_dev against optics.compute_raman (the restatement of reference optics.py:434-494 that the
    golden planes pin) on fresh temperatures and shift ratios: every element the same bits, for layer counts around the
    kernel's 30-layer passes; replacing the shift array or the table rebuilds the resident tables."""
    from picaso_amd import justdoit as jdi
    from picaso_amd import optics as px
    from picaso_amd.device import DeviceArray
    rng = np.random.default_rng(nlayer)
    opa = jdi.opannection(filename_db=DB)
    c, ji, dnu = og["in/raman_c"], og["in/raman_ji"], og["in/raman_deltanu"]
    opa.raman_db = {"c": c, "ji": ji, "deltanu": dnu}
    for trial in range(2):
        shifts = 1.0 + 0.3 * rng.standard_normal((opa.nwno, c.size))
        opa.raman_stellar_shifts = shifts
        tlayer = rng.uniform(40.0, 3000.0, nlayer)
        out = DeviceArray((nlayer, opa.nwno), opa.ctx)
        px.raman_oklopcic_device(opa, tlayer, out)
        want = np.minimum(px.compute_raman(opa.nwno, nlayer, opa.wno, shifts, tlayer, c, ji, dnu), 0.99999)
        assert np.array_equal(out.to_host(), want)
    opa.raman_db = {"c": c * 2.0, "ji": ji, "deltanu": dnu + 1.0}      # no Rayleigh transition left: all shifted
    px.raman_oklopcic_device(opa, tlayer, out)
    want = np.minimum(px.compute_raman(opa.nwno, nlayer, opa.wno, shifts, tlayer, c * 2.0, ji, dnu + 1.0), 0.99999)
    assert np.array_equal(out.to_host(), want, equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("raman", ["none", "oklopcic"])
def test_gpu_3d_facet_major_planes_equal_facet_fastest(og, raman, monkeypatch):
    """A 3-D spectrum without cloud: the default path (ONE fused gas + mixing launch over the tall atmosphere of all facets,
    facet-major planes, every facet a spectrum of its own in the batched solver launches) against the facet-fastest
    planes of ``compute_opacity_facets`` -- single spectrum and phase curve, per-facet temperatures AND abundances,
    bit for bit."""
    from picaso_amd import justdoit as jdi
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    if raman == "oklopcic":
        opa.raman_stellar_shifts = og["in/raman_shifts"]
        opa.raman_db = {"c": og["in/raman_c"], "ji": og["in/raman_ji"], "deltanu": og["in/raman_deltanu"]}
    ng, nt = 3, 2
    rng = np.random.default_rng(4)

    def profile(k):
        pr = {"pressure": og["in/plevel_bar"],
              "temperature": og["in/tlevel"][:, None, None] * (1.0 + 0.05 * k + 0.1 * rng.random((1, ng, nt)))}
        for m in ("H2", "He", "H2O", "CH4"):
            pr[m] = og["in/mix/" + m]
        pr["H2O"] = og["in/mix/H2O"][:, None, None] * (1.0 + 0.5 * rng.random((1, ng, nt)))
        return pr
    profs = [profile(k) for k in range(3)]

    def run():
        one = jdi.inputs()
        one.gravity(gravity=float(og["in/gravity"]))
        one.approx(raman=raman)
        one.phase_angle(0.7, num_gangle=ng, num_tangle=nt)
        one.atmosphere_3d(profs[0])
        a = one.spectrum(opa, calculation="reflected+thermal", dimension="3d")
        pc = jdi.inputs()
        pc.gravity(gravity=float(og["in/gravity"]))
        pc.approx(raman=raman)
        pc.phase_curve_geometry("reflected", [0.0, 1.1, 2.3], num_gangle=ng, num_tangle=nt)
        pc.atmosphere_4d(profs)
        return a, pc.phase_curve(opa)
    a_fm, c_fm = run()
    monkeypatch.setenv("PICASO_AMD_FACET_FASTEST", "1")
    a_ff, c_ff = run()
    for key in ("albedo", "thermal"):
        assert np.array_equal(a_fm[key], a_ff[key]), key
    for ph in c_ff:
        assert np.array_equal(c_fm[ph]["albedo"], c_ff[ph]["albedo"]), ph
