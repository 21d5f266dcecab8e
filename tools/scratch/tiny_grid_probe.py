"""spectrum() on tiny wavelength grids and two-level atmospheres: every per-wavelength output of a sub-grid must equal the
same entries of the full-grid spectrum bit for bit (the path is pointwise in wavelength), for Toon, SH4, transmission."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from picaso_amd import _lib
from picaso_amd import justdoit as jdi
from picaso_amd import optics as px
ctx = _lib.context(0)
NW = 130
wno_full = np.linspace(2000.0, 33333.0, NW)
temps, press = [100.0, 300.0, 700.0, 1500.0, 3000.0], [1e-6, 1e-4, 1e-2, 1e-1, 1.0, 10.0, 100.0, 500.0]
pt = [(i + 1, p, t) for i, (t, p) in enumerate((t, p) for t in temps for p in press)]
cia_t = [75.0, 500.0, 4000.0]
def make_opa(sel):
    wno = wno_full[sel]
    molecular = {m: {i: 10.0 ** (-24.0 + 2.0 * np.sin(wno / 2500.0 + k) + 0.4 * np.log10(p)) for (i, p, t) in pt} for k, m in enumerate(("H2O", "CH4"))}
    continuum = {pr: {t: 10.0 ** (-7.0 + np.cos(wno / 4000.0 + k)) for t in cia_t} for k, pr in enumerate(("H2H2", "H2He"))}
    ray = {m: 1e-27 * (wno / 1e4) ** 4 for m in ("H2", "He")}
    return px.RetrieveOpacities(wno, pt, molecular, continuum, cia_t, rayleigh_opa=ray, query_method="linear", ctx=ctx)
def case(nlevel, sh, sel, transit):
    plev = np.logspace(-4, 1, nlevel)
    prof = {"pressure": plev, "temperature": np.linspace(300.0, 1400.0, nlevel), "H2": np.full(nlevel, 0.84), "He": np.full(nlevel, 0.155),
            "H2O": np.full(nlevel, 1e-3), "CH4": np.full(nlevel, 5e-4)}
    c = jdi.inputs(); c.phase_angle(0); c.gravity(gravity=2500.0); c.atmosphere(df=prof)
    c.approx(**({"raman": "none", "rt_method": "SH", "stream": 4} if sh else {"raman": "none"}))
    if transit:
        c.star(relative_flux=(1.0 + 0.2 * np.cos(wno_full / 900.0))[sel], radius=6.9e10, semi_major=7.5e12)
        c.gravity(radius=7.1e9, mass=1.9e30)
    return c
bad = 0
for nlevel in (2, 3, 31):
    for sh in (False, True):
        for transit in (False, True):
            calc = "reflected+thermal" + ("+transmission" if transit else "")
            full_sel = slice(0, NW)
            ref = case(nlevel, sh, full_sel, transit).spectrum(make_opa(full_sel), calculation=calc)
            for k in (1, 2, 3, 63, 64, 65):
                sel = slice(0, k)
                try:
                    r = case(nlevel, sh, sel, transit).spectrum(make_opa(sel), calculation=calc)
                except Exception as e:
                    print("nlevel %d sh %d transit %d nwno %d: RAISED %s: %s" % (nlevel, sh, transit, k, type(e).__name__, str(e)[:120])); bad += 1; continue
                for key in ("albedo", "thermal", "transit_depth"):
                    if key in ref:
                        same = np.array_equal(r[key], ref[key][:k]); fin = bool(np.all(np.isfinite(r[key])))
                        if not (same and fin):
                            d = np.max(np.abs(r[key] - ref[key][:k]) / np.abs(ref[key][:k]))
                            print("nlevel %d sh %d transit %d nwno %d %s: equal %s finite %s maxrel %.2e" % (nlevel, sh, transit, k, key, same, fin, d)); bad += 1
print("problems:", bad)
