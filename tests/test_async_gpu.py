"""inputs.spectrum_async(): twelve different spectra -- Toon and SH4, cloud planes / patchy clouds / none, the symmetric disk
and a 6 x 6 disk at phase 0.8, level fluxes, 3-D, a star, transmission, full_output -- all enqueued before the first result
is read (more than ASYNC_DEPTH in flight, so block-table slots are reused while their previous spectrum is still pending),
results read in reverse order: every dictionary equals the plain spectrum() call bit for bit, whatever the order."""
import numpy as np
import pytest


def _setup():
    from picaso_amd import _lib
    from picaso_amd import justdoit as jdi
    from picaso_amd import optics as px
    ctx = _lib.context(0)
    nwno, nlevel = 700, 31
    wno = np.linspace(2000.0, 33333.0, nwno)
    temps, press = [100.0, 300.0, 700.0, 1500.0, 3000.0], [1e-6, 1e-4, 1e-2, 1e-1, 1.0, 10.0, 100.0, 500.0]
    pt = [(i + 1, p, t) for i, (t, p) in enumerate((t, p) for t in temps for p in press)]
    molecular = {m: {i: 10.0 ** (-24.0 + 2.0 * np.sin(wno / 2500.0 + k) + 0.4 * np.log10(p)) for (i, p, t) in pt}
                 for k, m in enumerate(("H2O", "CH4"))}
    cia_t = [75.0, 500.0, 4000.0]
    continuum = {pr: {t: 10.0 ** (-7.0 + np.cos(wno / 4000.0 + k)) for t in cia_t} for k, pr in enumerate(("H2H2", "H2He"))}
    ray = {m: 1e-27 * (wno / 1e4) ** 4 for m in ("H2", "He")}
    opa = px.RetrieveOpacities(wno, pt, molecular, continuum, cia_t, rayleigh_opa=ray, query_method="linear", ctx=ctx)
    plev = np.logspace(-6, 2, nlevel)

    def case(k):
        """(inputs, spectrum() keyword arguments)"""
        prof = {"pressure": plev, "temperature": (150.0 + 1200.0 * ((np.log10(plev) + 6) / 8) ** 2) * (1 + 0.02 * k),
                "H2": np.full(nlevel, 0.84), "He": np.full(nlevel, 0.155), "H2O": np.full(nlevel, 1e-3),
                "CH4": np.full(nlevel, 5e-4)}
        c = jdi.inputs()
        kw = {"calculation": "reflected+thermal"}
        if k == 6:                       # 3-D: 4 x 4 facets, per-facet temperatures
            pert = 1.0 + 0.1 * np.cos(np.arange(16).reshape(4, 4))
            c.phase_angle(np.pi / 3, num_gangle=4, num_tangle=4)
            c.gravity(gravity=2500.0)
            c.atmosphere_3d(dict(prof, temperature=prof["temperature"][:, None, None] * pert[None]))
            c.approx(raman="none")
            return c, dict(kw, dimension="3d")
        if k % 4 == 3:
            c.phase_angle(0.8, num_gangle=6, num_tangle=6)
        else:
            c.phase_angle(0)
        c.gravity(gravity=2500.0)
        c.atmosphere(df=prof)
        akw = {"raman": "none"}
        if k % 3 == 1:
            akw.update(rt_method="SH", stream=4)
        if k == 9:
            akw.update(get_lvl_flux=True)
        c.approx(**akw)
        if k % 2:
            shp = (nlevel - 1, nwno)
            opd = np.zeros(shp)
            opd[12 + k % 7:18 + k % 7] = 0.2
            c.clouds(df={"opd": opd, "w0": np.full(shp, 0.9), "g0": np.full(shp, 0.5)},
                     **(dict(do_holes=True, fhole=0.3, fthin_cld=0.1) if k == 5 else {}))
        if k == 8:                       # a star and the planet's size: flux ratios and the transit depth
            c.star(relative_flux=1.0 + 0.2 * np.cos(wno / 900.0), radius=6.9e10, semi_major=7.5e12)
            c.gravity(radius=7.1e9, mass=1.9e30)
            kw["calculation"] = "reflected+thermal+transmission"
        if k == 10:
            kw["full_output"] = True
        return c, kw
    return jdi, opa, [case(k) for k in range(12)]


def _same(a, b):
    keys = set(a) - {"full_output"}
    return keys == set(b) - {"full_output"} and all(np.array_equal(a[k], b[k]) for k in keys if isinstance(a[k], np.ndarray)) \
        and all(a[k] == b[k] for k in keys if isinstance(a[k], (float, str)))


@pytest.mark.gpu
def test_async_spectra_equal_plain_calls_in_any_order():
    jdi, opa, cases = _setup()
    single = [c.spectrum(opa, **kw) for c, kw in cases]
    assert len(cases) > jdi.ASYNC_DEPTH
    pending = [c.spectrum_async(opa, **kw) for c, kw in cases]            # twelve in flight, four slots
    outs = [None] * len(cases)
    for k in reversed(range(len(cases))):
        outs[k] = pending[k].result()
        assert pending[k].result() is outs[k] and pending[k].done()
    assert [k for k, (a, b) in enumerate(zip(outs, single)) if not _same(a, b)] == []
    # the pipelined loop of a retrieval: ask for sample i + 1, then read sample i
    prev, got = None, []
    for c, kw in cases:
        h = c.spectrum_async(opa, **kw)
        if prev is not None:
            got.append(prev.result())
        prev = h
    got.append(prev.result())
    assert [k for k, (a, b) in enumerate(zip(got, single)) if not _same(a, b)] == []
    # plain calls between pending ones use tables of their own
    h = cases[0][0].spectrum_async(opa, **cases[0][1])
    mid = cases[2][0].spectrum(opa, **cases[2][1])
    assert _same(mid, single[2]) and _same(h.result(), single[0])


@pytest.mark.gpu
def test_async_handle_that_is_dropped_and_errors_that_are_kept():
    jdi, opa, cases = _setup()
    c0, kw0 = cases[0]
    want = c0.spectrum(opa, **kw0)
    for _ in range(3 * jdi.ASYNC_DEPTH):          # handles nobody reads: their slots are finished on reuse
        c0.spectrum_async(opa, **kw0)
    assert _same(c0.spectrum_async(opa, **kw0).result(), want)
    bad = jdi.inputs()
    with pytest.raises(Exception, match="atmosphere"):
        bad.spectrum_async(opa)
    assert _same(c0.spectrum_async(opa, **kw0).result(), want)
    assert jdi.inputs.spectrum_async.__doc__ and jdi.picaso_async.__doc__


@pytest.mark.gpu
def test_opacity_stage_enqueued_ahead_equals_the_one_call():
    """spectrum() enqueues the opacity stage before it fills the legs' half of the job (picaso_toon_spectrum_phase 1, then 2);
    Options(one_phase=True) is the single C call of rounds 4-5: every case, every key, bit for bit -- and the async path too."""
    jdi, opa, cases = _setup()
    one = jdi.Options(one_phase=True)
    for k, (c, kw) in enumerate(cases):
        a = c.spectrum(opa, **kw)
        b = c.spectrum(opa, options=one, **kw)
        assert _same(a, b), k
        assert _same(c.spectrum_async(opa, options=one, **kw).result(), a), k
