"""Cloud-free 1-D spectra write 3 of the 13 compute_opacity planes (reference optics.py:26-431) and hand the solvers
aliases and constant planes for the rest (justdoit.picaso, `lean`): the values are the ones the full set holds --
cosb = cosb_og = ftau_cld = 0, ftau_ray = 1, gcos2 = 0.5, dtau_og = dtau, tau_og = tau, w0_og = w0 (optics.py:335-420
with TAUCLD = 0) -- so every result is bit-identical to PICASO_AMD_ALL_PLANES=1."""
import os

import numpy as np
import pytest

from helpers import GOLDEN

pytestmark = pytest.mark.gpu
DB = os.path.join(GOLDEN, "synthetic_opacities.db")


def _case(jdi, og, raman, de, lvl):
    case = jdi.inputs()
    case.phase_angle(0)
    case.gravity(radius=7.1e9, mass=1.9e30)
    prof = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"]}
    for k in ("H2", "He", "H2O", "CH4"):
        prof[k] = og["in/mix/" + k]
    case.atmosphere(df=prof)
    case.star(relative_flux=1.0 + 0.2 * np.cos(np.arange(len(og["in/wno"])) / 5.0), radius=6.9e10, semi_major=7.5e12)
    case.approx(raman=raman, delta_eddington=de, get_lvl_flux=lvl)
    return case


def _same(a, b, path=""):
    if isinstance(a, dict):
        assert list(a.keys()) == list(b.keys()), path
        for k in a:
            _same(a[k], b[k], path + "/" + str(k))
    elif isinstance(a, np.ndarray):
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), path
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for x, y in zip(a, b):
            _same(x, y, path)
    else:
        assert a == b or (a != a and b != b), (path, a, b)


@pytest.mark.parametrize("calc", ["reflected+thermal", "reflected", "thermal", "thermal+transmission"])
@pytest.mark.parametrize("raman,de,lvl", [("none", True, False), ("oklopcic", True, False), ("pollack", False, False),
                                          ("none", True, True)])
def test_lean_planes_equal_full_planes(monkeypatch, tmp_path, calc, raman, de, lvl):
    from picaso_amd import justdoit as jdi
    from picaso_amd import optics as px
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    if lvl and "transmission" in calc:
        pytest.skip("level fluxes belong to the reflected / thermal legs")
    if raman == "pollack":                                   # the reference's table, where the reference reads it
        w = np.sort(1e4 / og["in/wno"])
        (tmp_path / "opacities").mkdir()
        np.savetxt(tmp_path / "opacities" / "raman_fortran.txt", np.column_stack([w, 0.9 + 0.05 * np.cos(w)]))
        monkeypatch.setenv("picaso_refdata", str(tmp_path))
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    opa.raman_stellar_shifts = og["in/raman_shifts"]
    opa.raman_db = {"c": og["in/raman_c"], "ji": og["in/raman_ji"], "deltanu": og["in/raman_deltanu"]}
    calls = []
    real = px.compute_opacity_resident

    def spy(*a, **k):
        calls.append(None if k.get("want") is None else set(k["want"]))
        return real(*a, **k)
    monkeypatch.setattr(px, "compute_opacity_resident", spy)
    monkeypatch.delenv("PICASO_AMD_ALL_PLANES", raising=False)
    lean = _case(jdi, og, raman, de, lvl).spectrum(opa, calculation=calc, full_output=True)
    monkeypatch.setenv("PICASO_AMD_ALL_PLANES", "1")
    full = _case(jdi, og, raman, de, lvl).spectrum(opa, calculation=calc, full_output=True)
    _same(full, lean)
    assert len(calls) == 2 and len(calls[0]) <= 4 and len(calls[0]) < len(calls[1])     # the first call really was lean
    for key in ("albedo", "thermal", "transit_depth"):
        if key in lean:
            assert np.isfinite(lean[key]).all()


def test_lean_planes_not_used_with_clouds_or_test_mode(monkeypatch):
    from picaso_amd import justdoit as jdi
    from picaso_amd import optics as px
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    opa = jdi.opannection(filename_db=DB)
    monkeypatch.setenv("PICASO_AMD_NO_DRIVER", "1")       # the call-by-call path (the C driver makes the same choice:
    wants = []                                            # tests/test_driver_gpu.py::test_driver_lean_planes)
    real = px.compute_opacity_resident
    monkeypatch.setattr(px, "compute_opacity_resident", lambda *a, **k: (wants.append(k.get("want")), real(*a, **k))[1])
    case = _case(jdi, og, "none", True, False)
    case.clouds(df={"opd": og["in/cld_opd"], "w0": og["in/cld_w0"], "g0": og["in/cld_g0"]})
    cloudy = case.spectrum(opa, calculation="reflected")
    assert len(wants[-1]) == 8                   # the eleven reflected-light planes less tau, tau_og, gcos2 (re-derived)
    monkeypatch.setenv("PICASO_AMD_ALL_PLANES", "1")
    _same(case.spectrum(opa, calculation="reflected"), cloudy)
    assert len(wants[-1]) == 11
    monkeypatch.delenv("PICASO_AMD_ALL_PLANES")
    case = _case(jdi, og, "none", True, False)
    case.inputs["test_mode"] = "rayleigh"
    case.spectrum(opa, calculation="reflected")
    assert len(wants[-1]) == 11                  # a test mode: all eleven


@pytest.mark.parametrize("calc", ["reflected+thermal", "reflected", "thermal"])
def test_sh4_cloud_free_writes_two_planes(monkeypatch, calc):
    """rt_method='SH', stream 4, default forms, no cloud: dtau and w0 are written and the cloud-free SH launch
    (k_sh4_clear) solves them; reflected light agrees with the full-plane launch to the oracle's tolerance (not bit for
    bit: sh.hip), thermal emission -- the same kernel on the same two planes -- is bit-identical.  Other
    forms or layer fluxes keep the thirteen planes; a cloud with the default forms takes eight."""
    from picaso_amd import justdoit as jdi
    from picaso_amd import optics as px
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    opa = jdi.opannection(filename_db=DB)
    wants = []
    real = px.compute_opacity_resident
    monkeypatch.setattr(px, "compute_opacity_resident", lambda *a, **k: (wants.append(k.get("want")), real(*a, **k))[1])
    monkeypatch.setenv("PICASO_AMD_NO_DRIVER", "1")       # the call-by-call path (the C driver's choice: tests/test_driver_gpu.py)

    def case(**sh):
        c = _case(jdi, og, "none", True, False)
        c.approx(raman="none", rt_method="SH", stream=4, **sh)
        return c
    monkeypatch.delenv("PICASO_AMD_ALL_PLANES", raising=False)
    lean = case().spectrum(opa, calculation=calc)
    assert wants[-1] == {"dtau", "w0"}
    monkeypatch.setenv("PICASO_AMD_ALL_PLANES", "1")
    full = case().spectrum(opa, calculation=calc)
    assert wants[-1] is None
    monkeypatch.delenv("PICASO_AMD_ALL_PLANES")
    if "reflected" in calc:
        assert np.max(np.abs(lean["albedo"] - full["albedo"]) / np.abs(full["albedo"])) < 1e-9
        assert np.isfinite(lean["albedo"]).all() and (lean["albedo"] > 0).all()
    if "thermal" in calc:
        assert np.array_equal(lean["thermal"], full["thermal"])
    if calc == "reflected":
        # a cloud with the default forms: the eight planes the SH launch reads -- the level planes tau / tau_og are running
        # sums (running products of the beam exponentials in the kernel, round 5), cosb / gcos2 / w0_no_raman are read by
        # no SH solver -- within the rounding of the products of the full set's result
        c = case()
        c.clouds(df={"opd": og["in/cld_opd"], "w0": og["in/cld_w0"], "g0": og["in/cld_g0"]})
        eight = c.spectrum(opa, calculation=calc)
        assert wants[-1] == {"dtau", "w0", "cosb_og", "ftau_cld", "ftau_ray", "f_deltaM", "dtau_og", "w0_og"}
        monkeypatch.setenv("PICASO_AMD_ALL_PLANES", "1")
        thirteen = c.spectrum(opa, calculation=calc)
        assert wants[-1] is None
        monkeypatch.delenv("PICASO_AMD_ALL_PLANES")
        assert np.max(np.abs(eight["albedo"] - thirteen["albedo"]) / np.abs(thirteen["albedo"])) < 1e-11
        case(w_multi_form="OTHG").spectrum(opa, calculation=calc)
        assert wants[-1] is None
        case(calculate_fluxes="on").spectrum(opa, calculation=calc)
        assert wants[-1] is None
        case().spectrum(opa, calculation=calc, full_output=True)
        assert wants[-1] is None


def test_sh4_cloud_deck_uses_the_cloud_free_top(monkeypatch):
    """rt_method='SH', stream 4 with a cloud below layer 12: spectrum() states the cloud-free top to the SH launch
    (picaso_get_reflected_SH_top_dev; the statement is checked on the device here), the result agrees with the unsplit
    launch to 1e-9 and wavelength blocks (devices=[0, 0, 0]) reproduce it bit for bit."""
    from picaso_amd import justdoit as jdi
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    opa = jdi.opannection(filename_db=DB)
    nlayer = og["in/cld_opd"].shape[0]
    deck = 12
    cld = {k: np.array(og["in/cld_" + k]) for k in ("opd", "w0", "g0")}
    cld["opd"] = np.abs(cld["opd"]) + 0.05
    for k in ("opd", "g0"):
        cld[k][:deck] = 0.0

    def case():
        c = _case(jdi, og, "none", True, False)
        c.approx(raman="none", rt_method="SH", stream=4)
        c.clouds(df=cld)
        return c
    assert jdi._cloud_free_top(case().inputs, nlayer) == deck
    monkeypatch.setenv("PICASO_AMD_SH_CHECK_TOP", "1")
    split = case().spectrum(opa, calculation="reflected+thermal")
    blocks = case().spectrum(opa, calculation="reflected+thermal", devices=[0, 0, 0])
    monkeypatch.setenv("PICASO_AMD_SH_NO_TOP", "1")
    plain = case().spectrum(opa, calculation="reflected+thermal")
    assert np.array_equal(blocks["albedo"], split["albedo"]) and np.array_equal(blocks["thermal"], split["thermal"])
    assert not np.array_equal(split["albedo"], plain["albedo"])          # the split path did run
    assert np.max(np.abs(split["albedo"] - plain["albedo"]) / np.abs(plain["albedo"])) < 1e-9
    assert np.array_equal(split["thermal"], plain["thermal"])
