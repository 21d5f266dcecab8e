"""Transmission spectrum get_transit_1d: CPU oracle and HIP kernel against
tests/golden/transit.npz (outputs of the reference's own fluxes.get_transit_1d)."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, rel_err

CASES = ("a", "b", "two")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "transit.npz"))


def _args(g, c):
    nlevel, nwno = g[c + "/z"].size, g[c + "/dtau"].shape[1]
    return (g[c + "/z"], g[c + "/dz"], nlevel, nwno, float(g[c + "/rstar"]), g[c + "/mmw"],
            float(g[c + "/k_b"]), float(g[c + "/amu"]), g[c + "/plevel"], g[c + "/tlevel"],
            g[c + "/colden"], g[c + "/dtau"])


@pytest.mark.parametrize("c", CASES)
def test_oracle_transit(gold, oracle, c):
    # the transit depth is (zmin/Rs)^2 + a small atmospheric term: compare the atmospheric part too
    F = oracle.get_transit_1d(*_args(gold, c))
    base = (gold[c + "/z"].min() / float(gold[c + "/rstar"])) ** 2
    assert rel_err(F, gold[c + "/F"]) < 1e-13
    assert rel_err(F - base, gold[c + "/F"] - base) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES)
def test_gpu_transit(gold, c):
    from picaso_amd import fluxes
    F = fluxes.get_transit_1d(*_args(gold, c))
    base = (gold[c + "/z"].min() / float(gold[c + "/rstar"])) ** 2
    assert rel_err(F, gold[c + "/F"]) < 1e-13
    assert rel_err(F - base, gold[c + "/F"] - base) < 1e-9


@pytest.mark.gpu
def test_gpu_transit_vs_oracle_large(oracle):
    """90 layers x 5003 wavelengths (ragged last block), resident entry point."""
    import ctypes
    from picaso_amd import _lib, synthetic as syn
    from picaso_amd._lib import check, f64, load, ptr
    from picaso_amd.device import DeviceArray
    nlayer, nwno = 90, 5003
    sc = syn.make_scene(nlayer, nwno, seed=19)
    k_b, amu = 1.380649e-16, 1.66053906660e-24
    p, t = sc["plevel"], sc["tlevel"]
    mmw = np.full(nlayer, 2.3)
    H = k_b * 0.5 * (t[1:] + t[:-1]) / (mmw * amu * 2500.0)
    dzl = H * np.log(p[1:] / p[:-1])
    z = 7.0e9 + np.concatenate([np.cumsum(dzl[::-1])[::-1], [0.0]])
    dz = np.concatenate([dzl, [dzl[-1]]])
    colden = (p[1:] - p[:-1]) / 2500.0
    args = (z, dz, nlayer + 1, nwno, 6.96e10, mmw, k_b, amu, p, t, colden, sc["dtau_og"])
    Fo = oracle.get_transit_1d(*args)
    ctx = _lib.context()
    d = DeviceArray.from_host(sc["dtau_og"], ctx)
    out = DeviceArray((nwno,), ctx)
    check(load().picaso_get_transit_1d_dev(
        ctx, ptr(f64(z)), ptr(f64(dz)), ctypes.c_int(nlayer + 1), ctypes.c_int(nwno), ctypes.c_long(nwno),
        ctypes.c_double(6.96e10), ptr(f64(mmw)), ctypes.c_double(k_b), ctypes.c_double(amu), ptr(f64(p)),
        ptr(f64(t)), ptr(f64(colden)), ptr(d.addr), ptr(out.addr)), ctx)
    base = (z.min() / 6.96e10) ** 2
    assert rel_err(out.to_host() - base, Fo - base) < 1e-9


def _transit_case(og, jdi, **cloud_kw):
    case = jdi.inputs()
    case.phase_angle(0)
    case.gravity(radius=7.1e9, mass=1.9e30)
    case.star(relative_flux=np.ones(len(og["in/wno"])), radius=6.96e10, semi_major=7.5e11)
    prof = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"]}
    for k in ("H2", "He", "H2O", "CH4"):
        prof[k] = og["in/mix/" + k]
    case.atmosphere(df=prof)
    case.clouds(df={"opd": og["in/cld_opd"], "w0": og["in/cld_w0"], "g0": og["in/cld_g0"]}, **cloud_kw)
    case.approx(raman="none", p_reference=10)
    return case


def _oracle_depth(oracle, fo, dtau_og, wts, rstar=6.96e10):
    """The Gauss-point loop of justdoit.py:388-405 with the CPU oracle."""
    lv, ly = fo["level"], fo["layer"]
    nlevel, nwno = len(lv["z"]), dtau_og.shape[1]
    out = 0.0
    for ig, w in enumerate(wts):
        out = out + w * oracle.get_transit_1d(lv["z"], lv["dz"], nlevel, nwno, rstar, ly["mmw"], 1.380649e-16,
                                              1.66053906660e-24, lv["pressure"] * 1e6, lv["temperature"],
                                              ly["column_density"], np.ascontiguousarray(dtau_og[:, :, ig]))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["mono", "ck", "holes"])
def test_gpu_transmission_end_to_end(oracle, kind):
    """inputs.spectrum(calculation='transmission'): hydrostatic z/dz from ATMSETUP.get_altitude
    (pinned to the reference in test_altitude.py), DTAU_OG resident, correlated-k sum and
    patchy-cloud blend on the device; against the oracle on the run's own optical depths."""
    import test_ck_optics as tck
    from picaso_amd import justdoit as jdi
    og = np.load(os.path.join(os.path.dirname(__file__), "golden", "optics.npz"))
    if kind == "ck":
        ck = np.load(os.path.join(os.path.dirname(__file__), "golden", "ck.npz"))
        opa, wts = tck._ck_class(ck), ck["in/gauss_wts"]
    else:
        opa, wts = jdi.opannection(filename_db=tck.DB, query_method="linear"), np.array([1.0])
    fhole, fthin = 0.35, 0.2
    case = _transit_case(og, jdi, **(dict(do_holes=True, fhole=fhole, fthin_cld=fthin) if kind == "holes" else {}))
    out = case.spectrum(opa, calculation="transmission", full_output=True)
    fo = out["full_output"]
    nlayer, nwno = fo["taucld"].shape[:2]
    gas = fo["taugas"].reshape(nlayer, nwno, -1) + fo["tauray"].reshape(nlayer, nwno, -1)
    cld = og["in/cld_opd"][:, :, None]      # (full_output keeps the thinned deck when do_holes)
    assert np.all(np.diff(fo["level"]["z"]) < 0) and np.isfinite(out["transit_depth"]).all()
    want = _oracle_depth(oracle, fo, gas + cld, wts)
    if kind == "holes":
        want = (1.0 - fhole) * want + fhole * _oracle_depth(oracle, fo, gas + fthin * cld, wts)
    base = (fo["level"]["z"].min() / 6.96e10) ** 2
    assert rel_err(out["transit_depth"] - base, want - base) < 1e-9
    # the cloud deck and the molecular bands both show: depth varies with wavelength
    assert np.ptp(out["transit_depth"]) > 0


@pytest.mark.gpu
def test_gpu_transmission_needs_radii():
    import test_ck_optics as tck
    from picaso_amd import justdoit as jdi
    og = np.load(os.path.join(os.path.dirname(__file__), "golden", "optics.npz"))
    case = _transit_case(og, jdi)
    case.gravity(gravity=2500.0)
    with pytest.raises(Exception, match="radius"):
        case.spectrum(jdi.opannection(filename_db=tck.DB, query_method="linear"), calculation="transmission")
