"""Opacity pre-stage: (i) CPU -- oracle/optics_oracle.py against the reference-generated
tests/golden/optics.npz; (ii) GPU -- picaso_amd.optics (HBM-resident tables read from the
synthetic sqlite DB in the reference schema, k_opacity_gas + k_compute_opacity) against the same
fixture, and an end-to-end inputs.spectrum() run against the oracle chain."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, rel_err

NAMES = ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray", "gcos2", "dtau_og", "tau_og", "w0_og",
         "cosb_og", "w0_no_raman", "f_deltaM")
DB = os.path.join(GOLDEN, "synthetic_opacities.db")
# molecular weights as the reference's ATMSETUP.get_weights gives them (main-isotope masses), from the fixture
_G = np.load(os.path.join(GOLDEN, "optics.npz"))
WEIGHTS = {k: float(_G["in/weight/" + k]) for k in ("H2", "He", "H2O", "CH4")}
CASES = ("de1_s2_r2_tmnone", "de0_s2_r2_tmnone", "de1_s4_r0_tmnone", "de1_s2_r2_tmrayleigh",
         "de0_s2_r2_tmconstant_tau", "de1_s2_r1_tmnone")


@pytest.fixture(scope="module")
def pollack_table(tmp_path_factory):
    """The reference's raman_fortran.txt (carried as data in raman_pollack.npz) laid out where the
    reference looks for it: $picaso_refdata/opacities/raman_fortran.txt."""
    g = np.load(os.path.join(GOLDEN, "raman_pollack.npz"))
    root = tmp_path_factory.mktemp("refdata")
    os.makedirs(root / "opacities")
    np.savetxt(root / "opacities" / "raman_fortran.txt", np.column_stack([g["table/w"], g["table/f"]]),
               fmt="%.17g")
    old = os.environ.get("picaso_refdata")
    os.environ["picaso_refdata"] = str(root)
    yield g
    if old is None:
        del os.environ["picaso_refdata"]
    else:
        os.environ["picaso_refdata"] = old


def test_raman_pollack_matches_reference(pollack_table):
    from picaso_amd import optics as px
    g = pollack_table
    for name in ("vis", "wide"):
        # (pandas' fast float parser, which the reference reads the table with, is 1 ulp off here and there)
        np.testing.assert_allclose(px.raman_pollack(4, g[name + "/wave"]), g[name + "/factor"], rtol=1e-15)
    tab = (g["table/w"], g["table/f"])
    np.testing.assert_allclose(px.raman_pollack(2, g["vis/wave"], table=tab), g["vis/factor"][:2], rtol=1e-15)


def test_raman_pollack_without_table(monkeypatch):
    from picaso_amd import optics as px
    monkeypatch.delenv("picaso_refdata", raising=False)
    with pytest.raises(Exception, match="picaso_refdata"):
        px.raman_pollack(3, np.linspace(0.3, 1, 5))
    monkeypatch.setenv("picaso_refdata", "/nonexistent")
    with pytest.raises(Exception, match="not found"):
        px.raman_pollack(3, np.linspace(0.3, 1, 5))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "optics.npz"))


def _close(a, b, tol=1e-12):
    a, b = np.asarray(a), np.asarray(b)
    both_nan = np.isnan(a) & np.isnan(b)
    den = np.where(np.abs(b) > 0, np.abs(b), 1.0)
    err = np.where(both_nan, 0.0, np.abs(a - b) / den)
    return float(np.nanmax(err)) < tol and np.array_equal(np.isnan(a), np.isnan(b))


def _case_args(key):
    de, s, r, tm = key.split("_", 3)
    return bool(int(de[2])), int(s[1]), int(r[1]), (None if tm == "tmnone" else tm[2:])


def test_oracle_compute_opacity(gold, pollack_table):
    """oracle mixing algebra vs the reference: rebuild TAUGAS/TAURAY from the reference's own
    molecular/continuum planes, then compare all 13 outputs."""
    from oracle import optics_oracle as oo
    from picaso_amd import optics as px
    wno = gold["in/wno"]
    nlevel = len(gold["in/tlevel"])
    p = gold["in/plevel_bar"] * 1e6
    t = gold["in/tlevel"]
    mix = {k: 0.5 * (gold["in/mix/" + k][1:] + gold["in/mix/" + k][:-1]) for k in WEIGHTS}
    mmw_l = sum(gold["in/mix/" + k] * WEIGHTS[k] for k in WEIGHTS)
    mmw = 0.5 * (mmw_l[1:] + mmw_l[:-1])
    g = float(gold["in/gravity"])
    colden = gold["in/colden"]               # as the reference's ATMSETUP gives it (half-gravity end layers)
    tlayer = 0.5 * (t[1:] + t[:-1])
    plev = p / 1e6
    A = (tlayer / (t[:-1] * t[1:])) * (t[1:] * plev[1:] - t[:-1] * plev[:-1]) / (plev[1:] - plev[:-1])
    B = (tlayer / (t[:-1] * t[1:])) * (t[:-1] - t[1:]) / (plev[1:] - plev[:-1])
    COEF1 = 8.31446261815324 * 273.15 ** 2 * .5E5 * (A * (plev[1:] ** 2 - plev[:-1] ** 2) + B * (2. / 3.) * (
        plev[1:] ** 3 - plev[:-1] ** 3)) / (1.01325 ** 2 * (g / 100.0) * tlayer * mmw)
    import sqlite3
    conn = sqlite3.connect(DB)
    ray = {m: px._convert_array(b) for m, b in conn.execute("SELECT molecule, opacity FROM rayleigh")}
    conn.close()
    for qm in ("nearest", "linear"):
        taugas = np.zeros((nlevel - 1, len(wno)))
        for a, b in (("H2", "H2"), ("H2", "He"), ("H2", "CH4")):
            taugas += gold["%s/continuum_opa/%s" % (qm, a + b)] * (COEF1 * mix[a] * mix[b])[:, None]
        for m in ("H2O", "CH4", "H2"):
            taugas += gold["%s/molecular_opa/%s" % (qm, m)] * (colden * mix[m] / mmw)[:, None]
        tauray = np.zeros_like(taugas)
        for m in ("H2", "He", "CH4", "H2O"):
            tauray += ray[m][None, :] * (colden * mix[m] / mmw)[:, None]
        for key in CASES:
            de, s, r, tm = _case_args(key)
            if r == 0:
                rf = px.compute_raman(len(wno), nlevel - 1, wno, gold["in/raman_shifts"], tlayer,
                                      gold["in/raman_c"], gold["in/raman_ji"], gold["in/raman_deltanu"])
                rf = np.minimum(rf, 0.99999)
            elif r == 1:
                rf = np.minimum(px.raman_pollack(nlevel - 1, 1e4 / wno), 0.99999)
            else:
                rf = 0.99999
            out = oo.compute_opacity(taugas, tauray, gold["in/cld_opd"], gold["in/cld_w0"],
                                     gold["in/cld_g0"], rf, stream=s, delta_eddington=de, test_mode=tm)
            for nm, arr in zip(NAMES, out):
                assert _close(arr, gold["%s/%s/%s" % (qm, key, nm)], 1e-11), (qm, key, nm)


# ------------------------------------------------------------------------------------------------
def _bundle(gold, px_just, tm, de, stream_unused, raman):
    case = px_just.inputs()
    case.phase_angle(0)
    case.gravity(gravity=float(gold["in/gravity"]))
    prof = {"pressure": gold["in/plevel_bar"], "temperature": gold["in/tlevel"]}
    for k in ("H2", "He", "H2O", "CH4"):
        prof[k] = gold["in/mix/" + k]
    case.atmosphere(df=prof)
    case.clouds(df={"opd": gold["in/cld_opd"], "w0": gold["in/cld_w0"], "g0": gold["in/cld_g0"]})
    case.approx(raman=["oklopcic", "pollack", "none"][raman], delta_eddington=de)
    case.inputs["test_mode"] = tm
    return case


@pytest.mark.gpu
@pytest.mark.parametrize("qm", ["nearest", "linear"])
def test_gpu_compute_opacity(gold, qm, pollack_table):
    from picaso_amd import justdoit as jdi
    from picaso_amd import optics as px
    from picaso_amd.atmsetup import ATMSETUP
    opa = jdi.opannection(filename_db=DB, query_method=qm)
    assert opa.nwno == len(gold["in/wno"])
    opa.raman_stellar_shifts = gold["in/raman_shifts"]
    opa.raman_db = {"c": gold["in/raman_c"], "ji": gold["in/raman_ji"], "deltanu": gold["in/raman_deltanu"]}
    for key in CASES:
        de, s, r, tm = _case_args(key)
        case = _bundle(gold, jdi, tm, de, s, r)
        atm = ATMSETUP(case.inputs)
        atm.planet.gravity = case.inputs["planet"]["gravity"]
        atm.get_profile(); atm.get_mmw(); atm.get_altitude(); atm.get_column_density()
        atm.get_needed_continuum(opa.rayleigh_molecules, opa.avail_continuum)
        atm.get_clouds(opa.wno)
        atm.molecules = np.array([m for m in atm.molecules if m in opa.molecules])
        opa.get_opacities(atm)
        if key == CASES[0]:
            for m in ("H2O", "CH4", "H2"):
                assert _close(opa.molecular_opa[m], gold["%s/molecular_opa/%s" % (qm, m)], 1e-11), m
            for pr in ("H2H2", "H2He", "H2CH4"):
                assert _close(opa.continuum_opa[pr], gold["%s/continuum_opa/%s" % (qm, pr)], 1e-13), pr
        out = px.compute_opacity(atm, opa, ngauss=1, stream=s, delta_eddington=de, test_mode=tm, raman=r)
        assert len(out) == 13
        for nm, arr in zip(NAMES, out):
            assert arr.shape[2] == 1
            assert _close(arr[:, :, 0], gold["%s/%s/%s" % (qm, key, nm)], 1e-10), (qm, key, nm)
        # the default is ONE launch for gas stage + mixing (picaso_gas_compute_opacity_dev); the two launches with
        # TAUGAS / TAURAY through HBM give the same bits in all 13 planes
        os.environ["PICASO_AMD_UNFUSED_OPACITY"] = "1"
        try:
            two = px.compute_opacity(atm, opa, ngauss=1, stream=s, delta_eddington=de, test_mode=tm, raman=r)
        finally:
            del os.environ["PICASO_AMD_UNFUSED_OPACITY"]
        for nm, a1, a2 in zip(NAMES, out, two):
            assert np.array_equal(a1, a2), (qm, key, nm)


@pytest.mark.gpu
def test_gpu_spectrum_end_to_end(gold, oracle):
    """inputs.spectrum(): opacity tables -> compute_opacity -> Toon reflected + thermal -> disk
    integration entirely on the GPU, against the chain [reference compute_opacity planes from the
    fixture -> CPU oracle solvers]."""
    from picaso_amd import disco
    from picaso_amd import justdoit as jdi
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    case = _bundle(gold, jdi, None, True, 2, 2)
    case.surface_reflect(0.2)
    out = case.spectrum(opa, calculation="reflected+thermal", full_output=True)
    key = "linear/de1_s2_r2_tmnone"
    P = {nm: gold["%s/%s" % (key, nm)] for nm in NAMES}
    nlevel, nwno = P["tau"].shape
    g, gw, t, tw = disco.get_angles_1d(5)
    u0, u1, ct, _, _ = disco.compute_disco(5, 1, g, t, 0.0)
    x, _ = oracle.get_reflected_1d(nlevel, opa.wno, nwno, 5, 1, P["dtau"], P["tau"], P["w0"], P["cosb"],
                                   P["gcos2"], P["ftau_cld"], P["ftau_ray"], P["dtau_og"], P["tau_og"],
                                   P["w0_og"], P["cosb_og"], 0.2, u0, u1, 1.0, np.ones(nwno), 3, 0,
                                   1.0, -1.0, 2.0, -0.5, 1.0)
    alb = oracle.compress_disco(nwno, 1.0, x, gw, tw, np.ones(nwno))
    assert rel_err(out["albedo"], alb) < 1e-8
    f, _ = oracle.get_thermal_1d(nlevel, opa.wno, nwno, 5, 1, gold["in/tlevel"], P["dtau_og"],
                                 P["w0_no_raman"], P["cosb_og"], gold["in/plevel_bar"] * 1e6, u1,
                                 np.full(nwno, 0.2), 1, opa.wno * 0, 0)
    th = oracle.compress_thermal(nwno, f, gw, tw)
    assert rel_err(out["thermal"], th) < 1e-8
    wno = opa.wno
    assert np.isclose(out["bond_albedo"], np.trapezoid(x=1 / wno, y=alb) / np.trapezoid(x=1 / wno, y=alb * 0 + 1))
    assert np.isclose(out["effective_temperature"],
                      (np.trapezoid(x=1 / wno[::-1], y=th[::-1]) / 5.67e-5) ** 0.25)
    assert set(("wavenumber", "albedo", "bond_albedo", "fpfs_reflected", "thermal", "thermal_unit",
                "effective_temperature", "fpfs_thermal", "full_output")) <= set(out.keys())


@pytest.mark.gpu
def test_gpu_spectrum_sh4_end_to_end(gold, oracle):
    """approx(rt_method='SH', stream=4): opacity tables -> delta-M scaled planes (stream 4) ->
    SH4 reflected + thermal block sweeps -> disk integration, against [reference compute_opacity
    planes from the fixture -> CPU oracle SH solvers (LAPACK-style banded LU)]."""
    from picaso_amd import disco
    from picaso_amd import justdoit as jdi
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    opa.raman_stellar_shifts = gold["in/raman_shifts"]
    opa.raman_db = {"c": gold["in/raman_c"], "ji": gold["in/raman_ji"], "deltanu": gold["in/raman_deltanu"]}
    case = _bundle(gold, jdi, None, True, 4, 0)
    case.approx(raman="oklopcic", rt_method="SH", stream=4, delta_eddington=True, calculate_fluxes="on")
    out = case.spectrum(opa, calculation="reflected+thermal", full_output=True)
    key = "linear/de1_s4_r0_tmnone"
    P = {nm: gold["%s/%s" % (key, nm)] for nm in NAMES}
    nlevel, nwno = P["tau"].shape
    g, gw, t, tw = disco.get_angles_1d(5)
    u0, u1, ct, _, _ = disco.compute_disco(5, 1, g, t, 0.0)
    x, fl = oracle.get_reflected_SH(nlevel, nwno, 5, 1, P["dtau"], P["tau"], P["w0"], P["cosb"],
                                    P["ftau_cld"], P["ftau_ray"], P["f_deltaM"].copy(), P["dtau_og"],
                                    P["tau_og"], P["w0_og"], P["cosb_og"], 0.0, u0, u1, 1.0,
                                    np.ones(nwno), 0, 0, 0, 1, 1, 1, 1.0, -1.0, 2.0, -0.5, 1.0, 4, flx=1)
    alb = oracle.compress_disco(nwno, 1.0, x, gw, tw, np.ones(nwno))
    assert rel_err(out["albedo"], alb) < 1e-8
    from helpers import scale_err
    assert scale_err(out["full_output"]["flux_layers"], fl) < 1e-8    # calculate_fluxes='on' 
    f, _ = oracle.get_thermal_SH(nlevel, opa.wno, nwno, 5, 1, gold["in/tlevel"], P["dtau"], P["tau"],
                                 P["w0"], P["cosb"], P["dtau_og"], P["tau_og"], P["w0_og"],
                                 P["w0_no_raman"], P["cosb_og"], gold["in/plevel_bar"] * 1e6, u1,
                                 np.zeros(nwno), 4, 0)
    th = oracle.compress_thermal(nwno, f, gw, tw)
    assert rel_err(out["thermal"], th) < 1e-8


@pytest.mark.gpu
def test_gpu_symmetry_quadrant_equals_full_disk(gold):
    """phase_angle(symmetry=True): the 3x2 quadrant with the reference's doubled weights gives the
    full 6x4 disk's albedo and thermal flux for a horizontally uniform planet at full phase."""
    from picaso_amd import justdoit as jdi
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    outs = []
    for sym in (False, True):
        case = _bundle(gold, jdi, None, True, 2, 2)
        case.phase_angle(0, num_gangle=6, num_tangle=4, symmetry=sym)
        case.surface_reflect(0.1)
        outs.append(case.spectrum(opa, calculation="reflected+thermal", full_output=True))
    assert outs[1]["full_output"]["albedo_3d"].shape[:2] == (3, 2)
    assert rel_err(outs[1]["albedo"], outs[0]["albedo"]) < 1e-12
    assert rel_err(outs[1]["thermal"], outs[0]["thermal"]) < 1e-12


@pytest.mark.gpu
def test_gpu_spectrum_pollack_raman(gold, oracle, pollack_table):
    """approx(raman='pollack') end to end against the reference's compute_opacity(raman=1) planes."""
    from picaso_amd import disco
    from picaso_amd import justdoit as jdi
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    case = _bundle(gold, jdi, None, True, 2, 1)
    out = case.spectrum(opa, calculation="reflected")
    P = {nm: gold["linear/de1_s2_r1_tmnone/" + nm] for nm in NAMES}
    nlevel, nwno = P["tau"].shape
    g, gw, t, tw = disco.get_angles_1d(5)
    u0, u1, ct, _, _ = disco.compute_disco(5, 1, g, t, 0.0)
    x, _ = oracle.get_reflected_1d(nlevel, opa.wno, nwno, 5, 1, P["dtau"], P["tau"], P["w0"], P["cosb"],
                                   P["gcos2"], P["ftau_cld"], P["ftau_ray"], P["dtau_og"], P["tau_og"],
                                   P["w0_og"], P["cosb_og"], 0.0, u0, u1, 1.0, np.ones(nwno), 3, 0,
                                   1.0, -1.0, 2.0, -0.5, 1.0)
    assert rel_err(out["albedo"], oracle.compress_disco(nwno, 1.0, x, gw, tw, np.ones(nwno))) < 1e-8


def test_oracle_thinned_cloud_planes(gold):
    """compute_opacity(do_holes=True, fthin_cld=0.1) of the reference = the mixing with 0.1 x cloud opd."""
    from oracle import optics_oracle as oo
    key = "linear/de1_s2_r2_tmnone"
    dtau, taucld = gold[key + "/dtau_og"], gold["in/cld_opd"]
    # TAUGAS + TAURAY from the un-thinned reference planes: dtau_og - taucld; Rayleigh from ftau_ray / w0
    fray, w0c = gold[key + "/ftau_ray"], gold["in/cld_w0"]
    with np.errstate(invalid="ignore", divide="ignore"):
        tauray = np.where(fray < 1, fray * w0c * taucld / np.where(fray < 1, 1 - fray, 1.0), 0.0)
    tauray = np.where(taucld > 0, tauray, gold[key + "/w0_no_raman"] * dtau / 0.99999)
    taugas = dtau - tauray - taucld
    out = oo.compute_opacity(taugas, tauray, 0.1 * taucld, w0c, gold["in/cld_g0"], 0.99999, stream=2,
                             delta_eddington=True)
    for nm, arr in zip(NAMES, out):
        assert _close(arr, gold["linear/holes_fthin0.1/" + nm], 1e-9), nm


@pytest.mark.gpu
@pytest.mark.parametrize("qm", ["nearest", "linear"])
def test_gpu_thinned_cloud_planes(gold, qm):
    from picaso_amd import justdoit as jdi
    from picaso_amd import optics as px
    opa = jdi.opannection(filename_db=DB, query_method=qm)
    case = _bundle(gold, jdi, None, True, 2, 2)
    atm = jdi._setup_atmosphere(case.inputs, opa, opa.wno)
    opa.get_opacities(atm)
    out = px.compute_opacity(atm, opa, ngauss=1, stream=2, delta_eddington=True, test_mode=None, raman=2,
                             fthin_cld=0.1, do_holes=True)
    for nm, arr in zip(NAMES, out):
        assert _close(arr[:, :, 0], gold["%s/holes_fthin0.1/%s" % (qm, nm)], 1e-10), (qm, nm)


@pytest.mark.gpu
def test_gpu_single_leg_spectra_equal_the_combined_run(gold):
    """A thermal-only (reflected-only) spectrum asks the mixing kernel for 3 (11) of its 13 planes;
    the results are those of the combined run bit for bit."""
    from picaso_amd import justdoit as jdi
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    both = _bundle(gold, jdi, None, True, 2, 2).spectrum(opa, calculation="reflected+thermal")
    th = _bundle(gold, jdi, None, True, 2, 2).spectrum(opa, calculation="thermal")
    rf = _bundle(gold, jdi, None, True, 2, 2).spectrum(opa, calculation="reflected")
    assert np.array_equal(th["thermal"], both["thermal"]) and "albedo" not in th
    assert np.array_equal(rf["albedo"], both["albedo"]) and "thermal" not in rf


@pytest.mark.gpu
def test_gpu_overlapped_legs_stress(gold):
    """The thermal leg runs on a second stream next to the reflected leg; 150 back-to-back spectra of
    two alternating atmospheres must each reproduce their single-leg results bit for bit (a missing
    stream dependency would let one call's thermal kernel read the next call's planes)."""
    from picaso_amd import justdoit as jdi
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    cases, want = [], []
    for k in range(2):
        c = _bundle(gold, jdi, None, True, 2, 2)
        prof = dict(c.inputs["atmosphere"]["profile"])
        prof["temperature"] = np.asarray(prof["temperature"]) * (1.0 + 0.2 * k)
        prof["H2O"] = np.asarray(prof["H2O"]) * (1.0 + 3.0 * k)
        c.atmosphere(df=prof)
        cases.append(c)
        want.append((c.spectrum(opa, calculation="reflected")["albedo"], c.spectrum(opa, calculation="thermal")["thermal"]))
    assert not np.array_equal(want[0][1], want[1][1])
    for i in range(150):
        out = cases[i % 2].spectrum(opa, calculation="reflected+thermal")
        assert np.array_equal(out["albedo"], want[i % 2][0]) and np.array_equal(out["thermal"], want[i % 2][1]), i


@pytest.mark.gpu
def test_gpu_spectrum_level_fluxes_output_contract(gold, oracle):
    """approx(get_lvl_flux=True): picaso() returns disk-integrated level fluxes, not the raw
    (ng, nt, nlevel, nwno) arrays -- reflected: compress_disco(..., F0PI = 1) of every level
    (reference justdoit.py:536-548), thermal: the calc_type = 1 solve with dwno = wno*0
    (justdoit.py:322-327, 342), compress_thermal of every level times delta_wno (justdoit.py:575-580).
    Expected values: [reference compute_opacity planes from the fixture -> CPU oracle solvers with
    get_lvl_flux -> those reference lines restated with the oracle's compress_*]."""
    from picaso_amd import disco
    from picaso_amd import justdoit as jdi
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    case = _bundle(gold, jdi, None, True, 2, 2)
    case.surface_reflect(0.2)
    case.approx(raman="none", delta_eddington=True, get_lvl_flux=True)
    out = case.spectrum(opa, calculation="reflected+thermal", full_output=True)
    lev = out["full_output"]["level"]
    key = "linear/de1_s2_r2_tmnone"
    P = {nm: gold["%s/%s" % (key, nm)] for nm in NAMES}
    nlevel, nwno = P["tau"].shape
    g, gw, t, tw = disco.get_angles_1d(5)
    u0, u1, ct, _, _ = disco.compute_disco(5, 1, g, t, 0.0)
    x, lv = oracle.get_reflected_1d(nlevel, opa.wno, nwno, 5, 1, P["dtau"], P["tau"], P["w0"], P["cosb"],
                                    P["gcos2"], P["ftau_cld"], P["ftau_ray"], P["dtau_og"], P["tau_og"],
                                    P["w0_og"], P["cosb_og"], 0.2, u0, u1, 1.0, np.ones(nwno), 3, 0,
                                    1.0, -1.0, 2.0, -0.5, 1.0, get_toa_intensity=1, get_lvl_flux=1)
    names = ("flux_minus", "flux_plus", "flux_minus_mdpt", "flux_plus_mdpt")
    assert set(lev["reflected_fluxes"]) == set(names) and set(lev["thermal_fluxes"]) == set(names)
    for nm, data in zip(names, lv):
        want = np.array([oracle.compress_disco(nwno, 1.0, data[:, :, i, :], gw, tw, np.ones(nwno))
                         for i in range(nlevel)])
        got = lev["reflected_fluxes"][nm]
        assert got.shape == (nlevel, nwno)
        assert scale_err_rows(got, want) < 1e-8, nm
    f, tl = oracle.get_thermal_1d(nlevel, opa.wno, nwno, 5, 1, gold["in/tlevel"], P["dtau_og"],
                                  P["w0_no_raman"], P["cosb_og"], gold["in/plevel_bar"] * 1e6, u1,
                                  np.full(nwno, 0.2), 1, opa.wno * 0, 1)
    delta_wno = np.concatenate((np.diff(opa.wno), [np.diff(opa.wno)[-1]]))
    for nm, data in zip(names, tl):
        want = oracle.compress_thermal(nwno, data, gw, tw) * delta_wno
        got = lev["thermal_fluxes"][nm]
        assert got.shape == (nlevel, nwno)
        assert scale_err_rows(got, want) < 2e-7, nm          # thick-layer level fluxes: see helpers.lvl_excess
    assert rel_err(out["thermal"], oracle.compress_thermal(nwno, f, gw, tw)) < 1e-8


def scale_err_rows(got, want):
    """max |got - want| over the per-wavelength scale of the field"""
    scale = np.max(np.abs(want), axis=0, keepdims=True)
    scale = np.where(scale == 0, 1.0, scale)
    return float(np.max(np.abs(got - want) / scale))


@pytest.mark.gpu
@pytest.mark.parametrize("qm", ["linear", "nearest"])
def test_gpu_spectrum_config0_196_points_60_layers(oracle, qm):
    """BASELINE configs[0] at its stated shape: reflected-light spectrum on a 196-point opacity grid x 60
    layers through inputs.spectrum() (sqlite DB -> HBM tables -> gas stage -> mixing -> Toon solver ->
    disk integration), against [the reference's own compute_opacity planes for this DB and profile
    (tests/golden/optics_196x60.npz, generated by make_golden.py optics196) -> CPU oracle solver]."""
    from picaso_amd import disco
    from picaso_amd import justdoit as jdi
    g196 = np.load(os.path.join(GOLDEN, "optics_196x60.npz"))
    opa = jdi.opannection(filename_db=os.path.join(GOLDEN, "synthetic_opacities_196x60.db"), query_method=qm)
    assert opa.nwno == 196 and len(g196["in/tlevel"]) == 61
    case = _bundle(g196, jdi, None, True, 2, 2)
    case.surface_reflect(0.1)
    out = case.spectrum(opa, calculation="reflected+thermal", full_output=True)
    key = qm + "/de1_s2_r2_tmnone"          # get_opacities (linear) / get_opacities_nearest (optics.py:2310-2368)
    P = {nm: g196["%s/%s" % (key, nm)] for nm in NAMES}
    nlevel, nwno = P["tau"].shape
    assert (nlevel, nwno) == (61, 196)
    g, gw, t, tw = disco.get_angles_1d(5)
    u0, u1, ct, _, _ = disco.compute_disco(5, 1, g, t, 0.0)
    x, _ = oracle.get_reflected_1d(nlevel, opa.wno, nwno, 5, 1, P["dtau"], P["tau"], P["w0"], P["cosb"],
                                   P["gcos2"], P["ftau_cld"], P["ftau_ray"], P["dtau_og"], P["tau_og"],
                                   P["w0_og"], P["cosb_og"], 0.1, u0, u1, 1.0, np.ones(nwno), 3, 0,
                                   1.0, -1.0, 2.0, -0.5, 1.0)
    assert rel_err(out["albedo"], oracle.compress_disco(nwno, 1.0, x, gw, tw, np.ones(nwno))) < 1e-8
    f, _ = oracle.get_thermal_1d(nlevel, opa.wno, nwno, 5, 1, g196["in/tlevel"], P["dtau_og"], P["w0_no_raman"],
                                 P["cosb_og"], g196["in/plevel_bar"] * 1e6, u1, np.full(nwno, 0.1), 1, opa.wno * 0, 0)
    assert rel_err(out["thermal"], oracle.compress_thermal(nwno, f, gw, tw)) < 1e-8


@pytest.mark.gpu
def test_gpu_spectrum_with_a_star_file_and_oklopcic_raman(gold, tmp_path):
    """The reference's call sequence for Raman scattering after Oklopcic+2016: approx(raman='oklopcic'), then
    star(opa, filename=..., w_unit=..., f_unit=...) -- which bins the star and leaves the shifted / unshifted ratios
    on the opacity object (justdoit.py:1833-1842) -- then spectrum(): compute_raman on the device from those ratios."""
    from picaso_amd import justdoit as jdi
    from picaso_amd import optics as px
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    opa.raman_db = {"c": gold["in/raman_c"], "ji": gold["in/raman_ji"], "deltanu": gold["in/raman_deltanu"]}
    wave_um = np.linspace(0.05, 300.0, 200000)
    flux = 2e6 * np.exp(-((np.log(wave_um) - np.log(0.5)) / 1.2) ** 2) * (1.0 + 0.2 * np.sin(wave_um * 40.0)) + 1.0
    star_file = tmp_path / "star.txt"
    np.savetxt(star_file, np.column_stack([wave_um, flux]))
    outs = {}
    for raman in ("oklopcic", "none"):
        case = _bundle(gold, jdi, None, True, 2, 2)
        case.approx(raman=raman)
        case.gravity(radius=7.0e9, mass=1.9e30)
        case.star(opa, filename=str(star_file), w_unit="um", f_unit="erg/cm2/s/um", radius=1, radius_unit="R_sun",
                  semi_major=0.05, semi_major_unit="au")
        assert opa.relative_flux.shape == (opa.nwno,) and np.all(opa.relative_flux > 0)
        outs[raman] = case.spectrum(opa, calculation="reflected", full_output=True)
    assert opa.raman_stellar_shifts.shape == (opa.nwno, len(gold["in/raman_c"]))
    a, b = outs["oklopcic"], outs["none"]
    assert np.isfinite(a["albedo"]).all() and np.isfinite(a["fpfs_reflected"]).all()
    assert not np.array_equal(a["albedo"], b["albedo"])
    # the plane the mixing saw = min(compute_raman(host restatement), 0.99999): through w0 of the full output
    from picaso_amd.atmsetup import ATMSETUP
    rf = np.minimum(px.compute_raman(opa.nwno, len(gold["in/tlevel"]) - 1, opa.wno, opa.raman_stellar_shifts,
                                     0.5 * (gold["in/tlevel"][1:] + gold["in/tlevel"][:-1]), gold["in/raman_c"],
                                     gold["in/raman_ji"], gold["in/raman_deltanu"]), 0.99999)
    assert np.isfinite(rf).all() and rf.min() > 0
