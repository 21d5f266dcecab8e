"""The product call's paths against each other over a seeded sample of its switches -- solver family, cloud form (none, planes
on the opacity grid, tables on a grid of their own, patchy), legs, star, disk geometry, delta-Eddington, level
fluxes, surface --: spectrum() as it runs by default (C driver, opacity stage enqueued ahead), the single C call
(Options(one_phase=True)), the call-by-call path (Options(no_driver=True)), spectrum_async() and spectrum_batch(): every key of
every dictionary bit for bit.  Round 5 ran matrices like this one as one-off probes; this one stays."""
import itertools
import os

import numpy as np
import pytest


def _world():
    from picaso_amd import _lib
    from picaso_amd import justdoit as jdi
    from picaso_amd import optics as px
    ctx = _lib.context(0)
    nwno, nlevel = 420, 27
    wno = np.linspace(2500.0, 30000.0, nwno)
    temps, press = [100.0, 300.0, 700.0, 1500.0, 3000.0], [1e-6, 1e-4, 1e-2, 1e-1, 1.0, 10.0, 100.0, 500.0]
    pt = [(i + 1, p, t) for i, (t, p) in enumerate((t, p) for t in temps for p in press)]
    molecular = {m: {i: 10.0 ** (-24.0 + 2.0 * np.sin(wno / 2500.0 + k) + 0.4 * np.log10(p)) for (i, p, t) in pt}
                 for k, m in enumerate(("H2O", "CH4"))}
    cia_t = [75.0, 500.0, 4000.0]
    continuum = {pr: {t: 10.0 ** (-7.0 + np.cos(wno / 4000.0 + k)) for t in cia_t} for k, pr in enumerate(("H2H2", "H2He"))}
    ray = {m: 1e-27 * (wno / 1e4) ** 4 for m in ("H2", "He")}
    opa = px.RetrieveOpacities(wno, pt, molecular, continuum, cia_t, rayleigh_opa=ray, query_method="linear", ctx=ctx)
    plev = np.logspace(-6, 2, nlevel)
    return jdi, opa, wno, nwno, nlevel, plev


def _case(jdi, wno, nwno, nlevel, plev, rng, k):
    prof = {"pressure": plev, "temperature": (150.0 + 1200.0 * ((np.log10(plev) + 6) / 8) ** 2) * (1 + 0.03 * rng.random()),
            "H2": np.full(nlevel, 0.84), "He": np.full(nlevel, 0.155), "H2O": np.full(nlevel, 1e-3 * (1 + rng.random())),
            "CH4": np.full(nlevel, 5e-4)}
    c = jdi.inputs()
    geo = int(rng.integers(0, 3))
    if geo == 0:
        c.phase_angle(0)
    elif geo == 1:
        c.phase_angle(0, num_gangle=int(rng.choice([6, 8])))
    else:
        c.phase_angle(float(rng.uniform(0.2, 1.5)), num_gangle=int(rng.choice([2, 4, 6])), num_tangle=int(rng.choice([2, 4])))
    c.gravity(gravity=2500.0)
    c.atmosphere(df=prof)
    sh = bool(rng.integers(0, 3) == 0)
    raman = "none"                      # (the Raman forms need the reference's data files: tests/test_optics.py, test_ck_optics.py)
    akw = {"raman": raman, "delta_eddington": bool(rng.integers(0, 2))}
    if sh:
        akw.update(rt_method="SH", stream=int(rng.choice([2, 4])))
    lvl = (not sh) and rng.integers(0, 8) == 0
    if lvl:
        akw.update(get_lvl_flux=True)
    c.approx(**akw)
    cloud = int(rng.integers(0, 3))
    nl = nlevel - 1
    if cloud == 1:                      # planes on the opacity grid
        opd = np.zeros((nl, nwno))
        top = int(rng.integers(5, 15))
        opd[top:top + 5] = rng.uniform(0.05, 1.0)
        c.clouds(df={"opd": opd, "w0": np.full((nl, nwno), rng.uniform(0.5, 0.99)), "g0": np.full((nl, nwno), rng.uniform(0.1, 0.8))},
                 **(dict(do_holes=True, fhole=0.3, fthin_cld=0.1) if (not sh and rng.integers(0, 3) == 0) else {}))
    elif cloud == 2:                    # tables on a wavenumber grid of their own
        box = np.zeros((nl, 60))
        box[10:16] = rng.uniform(0.1, 0.5)
        c.clouds(df={"opd": box, "w0": np.where(box > 0, 0.95, 0.0), "g0": np.where(box > 0, 0.6, 0.0)},
                 wavenumber=np.linspace(wno[0], wno[-1], 60))
    if rng.integers(0, 3) == 0:
        c.surface_reflect(float(rng.uniform(0.05, 0.5)))
    calc = str(rng.choice(["reflected", "thermal", "reflected+thermal", "reflected+thermal"]))
    if rng.integers(0, 3) == 0:
        c.star(relative_flux=1.0 + 0.2 * np.cos(wno / (800.0 + 200.0 * rng.random())), radius=6.9e10, semi_major=7.5e12)
        c.gravity(radius=7.1e9, mass=1.9e30)
        if "reflected" in calc and not sh and not lvl and rng.integers(0, 2):
            calc += "+transmission"
    return c, calc, raman


def _same(a, b):
    ka, kb = set(a) - {"full_output"}, set(b) - {"full_output"}
    if ka != kb:
        return False
    for k in ka:
        if isinstance(a[k], np.ndarray):
            if not np.array_equal(a[k], b[k]):
                return False
        elif isinstance(a[k], (float, str, list)) and not (a[k] == b[k]):
            return False
    return True


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(3))
def test_product_paths_agree_bit_for_bit(block):
    jdi, opa, wno, nwno, nlevel, plev = _world()
    rng = np.random.default_rng(9100 + block + 7919 * int(os.environ.get("PICASO_FUZZ_OFFSET", "0")))     # soaks: other offsets
    one, nod = jdi.Options(one_phase=True), jdi.Options(no_driver=True)
    cases, ref, bad = [], [], []
    for k in range(24):
        c, calc, raman = _case(jdi, wno, nwno, nlevel, plev, rng, k)
        a = c.spectrum(opa, calculation=calc)
        for name, out in (("one_phase", c.spectrum(opa, calculation=calc, options=one)),
                          ("no_driver", c.spectrum(opa, calculation=calc, options=nod)),
                          ("async", c.spectrum_async(opa, calculation=calc).result()),
                          ("async one_phase", c.spectrum_async(opa, calculation=calc, options=one).result())):
            if not _same(a, out):
                bad.append((block, k, name, calc))
        assert all(np.all(np.isfinite(v)) for v in a.values() if isinstance(v, np.ndarray) and v.dtype == np.float64), (block, k)
        cases.append((c, calc))
        ref.append(a)
    assert bad == []
    # the same cases as a batch (members grouped by calculation: spectrum_batch takes one string)
    for calc, grp in itertools.groupby(sorted(range(len(cases)), key=lambda i: cases[i][1]), key=lambda i: cases[i][1]):
        idx = list(grp)
        outs = jdi.spectrum_batch([cases[i][0] for i in idx], opa, calculation=calc, batch_size=3)
        assert [i for i, o in zip(idx, outs) if not _same(ref[i], o)] == [], calc
