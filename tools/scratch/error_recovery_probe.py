"""A failing call must not poison the next one: error -> good call sequences on one opacity object."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from picaso_amd import _lib
from picaso_amd import justdoit as jdi
from picaso_amd import optics as px
ctx = _lib.context(0)
nwno, nlevel = 6000, 61
wno = np.linspace(2000.0, 33333.0, nwno)
temps, press = [100.0, 300.0, 700.0, 1500.0, 3000.0], [1e-6, 1e-4, 1e-2, 1e-1, 1.0, 10.0, 100.0, 500.0]
pt = [(i + 1, p, t) for i, (t, p) in enumerate((t, p) for t in temps for p in press)]
molecular = {m: {i: 10.0 ** (-24.0 + 2.0 * np.sin(wno / 2500.0 + k) + 0.4 * np.log10(p)) for (i, p, t) in pt} for k, m in enumerate(("H2O", "CH4"))}
cia_t = [75.0, 500.0, 4000.0]
continuum = {pr: {t: 10.0 ** (-7.0 + np.cos(wno / 4000.0 + k)) for t in cia_t} for k, pr in enumerate(("H2H2", "H2He"))}
ray = {m: 1e-27 * (wno / 1e4) ** 4 for m in ("H2", "He")}
opa = px.RetrieveOpacities(wno, pt, molecular, continuum, cia_t, rayleigh_opa=ray, query_method="linear", ctx=ctx)
plev = np.logspace(-6, 2, nlevel)
prof = {"pressure": plev, "temperature": 150.0 + 1200.0 * ((np.log10(plev) + 6) / 8) ** 2, "H2": np.full(nlevel, 0.84),
        "He": np.full(nlevel, 0.155), "H2O": np.full(nlevel, 1e-3), "CH4": np.full(nlevel, 5e-4)}
def base(**akw):
    c = jdi.inputs(); c.phase_angle(0); c.gravity(gravity=2500.0); c.atmosphere(df=prof); c.approx(raman="none", **akw); return c
good = base()
ref = good.spectrum(opa, calculation="reflected+thermal")
def check(tag):
    r = good.spectrum(opa, calculation="reflected+thermal")
    ok = np.array_equal(r["albedo"], ref["albedo"]) and np.array_equal(r["thermal"], ref["thermal"])
    print("   after %-40s next call %s" % (tag, "OK" if ok else "DIFFERS"), flush=True)
def attempt(tag, fn):
    try:
        fn(); print("%-44s did not raise" % tag, flush=True)
    except BaseException as e:
        print("%-44s raised %s: %s" % (tag, type(e).__name__, str(e)[:90]), flush=True)
    check(tag)
# 1. transmission without a star: raises after the other legs were enqueued
attempt("transmission without star", lambda: base().spectrum(opa, calculation="reflected+thermal+transmission"))
# 2. cloud table of the wrong shape
def badcloud():
    c = base(); c.clouds(df={"opd": np.zeros((10, 7)), "w0": np.zeros((10, 7)), "g0": np.zeros((10, 7))}); c.spectrum(opa, calculation="reflected")
attempt("cloud table of the wrong shape", badcloud)
# 3. SH thermal flx=1 (broken upstream: clean error here)
attempt("unknown calculation string", lambda: base().spectrum(opa, calculation="emission"))
# 4. NaN temperature: propagates
def nant():
    c = jdi.inputs(); c.phase_angle(0); c.gravity(gravity=2500.0)
    p2 = dict(prof, temperature=prof["temperature"].copy()); p2["temperature"][10] = np.nan
    c.atmosphere(df=p2); c.approx(raman="none"); r = c.spectrum(opa, calculation="reflected+thermal"); print("      nan count", int(np.isnan(r["thermal"]).sum()))
attempt("NaN in the temperature profile", nant)
# 5. KeyboardInterrupt-like exception thrown between enqueue and finish (defer)
def interrupted():
    s = jdi.picaso(good.inputs, opa, calculation="reflected+thermal", defer=True)
    raise KeyboardInterrupt("between enqueue and finish")
attempt("interrupt between enqueue and finish", interrupted)
# 6. 3-D without atmosphere_3d
attempt("dimension='3d' on a 1-D case", lambda: base().spectrum(opa, calculation="reflected", dimension="3d"))
# 7. devices with a bad device index
attempt("devices=[0, 99]", lambda: base().spectrum(opa, calculation="reflected", devices=[0, 99]))
check("everything")
