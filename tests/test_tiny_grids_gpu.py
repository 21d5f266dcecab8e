"""spectrum() on tiny wavelength grids and two-level atmospheres, all three legs in one call: every per-wavelength output of
a sub-grid equals the same entries of the full-grid spectrum bit for bit (the path is pointwise in wavelength; reference
justdoit.py:236-405).  Found with this probe in round 5: a cloud-free SH spectrum with the transmission leg raised
KeyError 'dtau_og' (the lean SH plane set did not name the plane the transit kernel reads)."""
import numpy as np
import pytest

NW = 130
WNO = np.linspace(2000.0, 33333.0, NW)
TEMPS, PRESS = [100.0, 300.0, 700.0, 1500.0, 3000.0], [1e-6, 1e-4, 1e-2, 1e-1, 1.0, 10.0, 100.0, 500.0]
PT = [(i + 1, p, t) for i, (t, p) in enumerate((t, p) for t in TEMPS for p in PRESS)]
CIA_T = [75.0, 500.0, 4000.0]


def _opa(k):
    from picaso_amd import _lib
    from picaso_amd import optics as px
    wno = WNO[:k]
    molecular = {m: {i: 10.0 ** (-24.0 + 2.0 * np.sin(wno / 2500.0 + j) + 0.4 * np.log10(p)) for (i, p, t) in PT}
                 for j, m in enumerate(("H2O", "CH4"))}
    continuum = {pr: {t: 10.0 ** (-7.0 + np.cos(wno / 4000.0 + j)) for t in CIA_T} for j, pr in enumerate(("H2H2", "H2He"))}
    ray = {m: 1e-27 * (wno / 1e4) ** 4 for m in ("H2", "He")}
    return px.RetrieveOpacities(wno, PT, molecular, continuum, CIA_T, rayleigh_opa=ray, query_method="linear",
                                ctx=_lib.context(0))


def _case(nlevel, sh, k):
    from picaso_amd import justdoit as jdi
    plev = np.logspace(-4, 1, nlevel)
    prof = {"pressure": plev, "temperature": np.linspace(300.0, 1400.0, nlevel), "H2": np.full(nlevel, 0.84),
            "He": np.full(nlevel, 0.155), "H2O": np.full(nlevel, 1e-3), "CH4": np.full(nlevel, 5e-4)}
    c = jdi.inputs()
    c.phase_angle(0)
    c.atmosphere(df=prof)
    c.approx(**({"raman": "none", "rt_method": "SH", "stream": 4} if sh else {"raman": "none"}))
    c.star(relative_flux=(1.0 + 0.2 * np.cos(WNO / 900.0))[:k], radius=6.9e10, semi_major=7.5e12)
    c.gravity(radius=7.1e9, mass=1.9e30)
    return c


@pytest.mark.gpu
@pytest.mark.parametrize("sh", [False, True])
@pytest.mark.parametrize("nlevel", [2, 3, 31])
def test_sub_grids_equal_the_full_grid_column_for_column(nlevel, sh):
    calc = "reflected+thermal+transmission"
    with np.errstate(invalid="ignore", divide="ignore"):       # one-point grids: the spectrum-wide integrals are 0 / 0
        ref = _case(nlevel, sh, NW).spectrum(_opa(NW), calculation=calc)
        for k in (1, 2, 3, 63, 64, 65):
            r = _case(nlevel, sh, k).spectrum(_opa(k), calculation=calc)
            for key in ("albedo", "thermal", "transit_depth"):
                assert np.all(np.isfinite(r[key])), (k, key)
                assert np.array_equal(r[key], ref[key][:k]), (k, key)
