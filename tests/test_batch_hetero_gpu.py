"""spectrum_batch over a HETEROGENEOUS list -- Toon and SH4, cloud tables / patchy clouds / none, the symmetric disk and a 6 x 6
disk at phase 0.8 (36 angles: more than the batched launch carries; it used to raise "at most 8 disk angles"), level
fluxes -- and spectrum(devices=[0, 0, 0]) for each case: every output equals the plain spectrum() call bit for bit."""
import numpy as np
import pytest


@pytest.mark.gpu
def test_heterogeneous_batch_and_blocks_equal_single_calls():
    from picaso_amd import _lib
    from picaso_amd import justdoit as jdi
    from picaso_amd import optics as px
    ctx = _lib.context(0)
    nwno, nlevel = 700, 31
    wno = np.linspace(2000.0, 33333.0, nwno)
    temps, press = [100.0, 300.0, 700.0, 1500.0, 3000.0], [1e-6, 1e-4, 1e-2, 1e-1, 1.0, 10.0, 100.0, 500.0]
    pt = [(i + 1, p, t) for i, (t, p) in enumerate((t, p) for t in temps for p in press)]
    molecular = {m: {i: 10.0 ** (-24.0 + 2.0 * np.sin(wno / 2500.0 + k) + 0.4 * np.log10(p)) for (i, p, t) in pt} for k, m in enumerate(("H2O", "CH4"))}
    cia_t = [75.0, 500.0, 4000.0]
    continuum = {pr: {t: 10.0 ** (-7.0 + np.cos(wno / 4000.0 + k)) for t in cia_t} for k, pr in enumerate(("H2H2", "H2He"))}
    ray = {m: 1e-27 * (wno / 1e4) ** 4 for m in ("H2", "He")}
    opa = px.RetrieveOpacities(wno, pt, molecular, continuum, cia_t, rayleigh_opa=ray, query_method="linear", ctx=ctx)
    plev = np.logspace(-6, 2, nlevel)
    def case(k):
        prof = {"pressure": plev, "temperature": (150.0 + 1200.0 * ((np.log10(plev) + 6) / 8) ** 2) * (1 + 0.02 * k), "H2": np.full(nlevel, 0.84),
                "He": np.full(nlevel, 0.155), "H2O": np.full(nlevel, 1e-3), "CH4": np.full(nlevel, 5e-4)}
        c = jdi.inputs()
        if k % 4 == 3: c.phase_angle(0.8, num_gangle=6, num_tangle=6)
        else: c.phase_angle(0)
        c.gravity(gravity=2500.0); c.atmosphere(df=prof)
        akw = {"raman": "none"}
        if k % 3 == 1: akw.update(rt_method="SH", stream=4)
        if k % 5 == 4: akw.update(get_lvl_flux=True)
        c.approx(**akw)
        if k % 2:
            shp = (nlevel - 1, nwno); opd = np.zeros(shp); opd[12 + k % 7:18 + k % 7] = 0.2
            c.clouds(df={"opd": opd, "w0": np.full(shp, 0.9), "g0": np.full(shp, 0.5)}, **(dict(do_holes=True, fhole=0.3, fthin_cld=0.1) if k % 6 == 5 else {}))
        return c
    cases = [case(k) for k in range(12)]
    calc = "reflected+thermal"
    single = [c.spectrum(opa, calculation=calc) for c in cases]

    def same(a, b):
        return set(a) == set(b) and all(np.array_equal(a[k], b[k]) for k in a if isinstance(a[k], np.ndarray))
    for bs in (3, 12):
        out = jdi.spectrum_batch(cases, opa, calculation=calc, batch_size=bs)
        assert [k for k, (a, b) in enumerate(zip(out, single)) if not same(a, b)] == [], "batch_size=%d" % bs
    for k in (0, 1, 3, 4, 5):
        assert same(cases[k].spectrum(opa, calculation=calc, devices=[0, 0, 0]), single[k]), k
