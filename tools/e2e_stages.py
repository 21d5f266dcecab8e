#!/usr/bin/env python
"""Where one spectrum() call of the 1-D driver path spends its host time: wall-clock per stage (wrappers around the
stage functions, no profiler), median of 200 calls.  Run on the GPU box: NWNO=100000 python tools/e2e_stages.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("WARM", "5")
from picaso_amd import justdoit as jdi, optics as px, driver as drv, atmsetup

acc = {}


def wrap(mod, name, label=None):
    f = getattr(mod, name)
    label = label or name

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc.setdefault(label, []).append(time.perf_counter() - t0)
    setattr(mod, name, g)


wrap(jdi, "_setup_atmosphere")
wrap(px.RetrieveOpacities, "get_opacities")
wrap(px, "_layer_factors")
wrap(drv, "make_job")
wrap(drv, "enqueue")
wrap(drv, "collect")
wrap(jdi, "_post_reflected")
wrap(jdi, "_post_thermal")
wrap(jdi, "_resident_vector")
wrap(jdi, "_picaso_driver")
wrap(jdi, "picaso")
wrap(jdi, "_opacity_shards")
from picaso_amd import onecall
wrap(onecall, "finish")
wrap(onecall, "prepare")
for _n in ("_setup_atmosphere", "_call_state", "_cloud_inputs", "_fill_block", "_post_reflected", "_post_thermal",
           "_post_final", "_plane_set", "_prepared", "_in_scope"):
    wrap(onecall, _n, "onecall." + _n)
devs = [int(x) for x in os.environ["DEVICES"].split(",")] if os.environ.get("DEVICES") else None

# the scene of tools/e2e_1d_time.py
from picaso_amd import _lib
nwno, nlevel = int(os.environ.get("NWNO", "100000")), 91
ctx = _lib.context(0)
wno = np.linspace(2000.0, 33333.0, nwno)
temps, press = [100.0, 300.0, 700.0, 1500.0, 3000.0], [1e-6, 1e-4, 1e-2, 1e-1, 1.0, 10.0, 100.0, 500.0]
pt = [(i + 1, p, t) for i, (t, p) in enumerate((t, p) for t in temps for p in press)]
mols = ["H2O", "CH4", "CO", "NH3", "H2"]
molecular = {m: {i: 10.0 ** (-24.0 + 2.0 * np.sin(wno / 2500.0 + k) + 0.4 * np.log10(p) + 0.8 * np.log10(t / 300.0))
                 for (i, p, t) in pt} for k, m in enumerate(mols)}
cia_t = [75.0, 200.0, 500.0, 1000.0, 2000.0, 4000.0]
continuum = {pr: {t: 10.0 ** (-7.0 + np.cos(wno / 4000.0 + k) + 0.3 * np.log10(t / 300.0)) for t in cia_t}
             for k, pr in enumerate(("H2H2", "H2He"))}
ray = {m: 1e-27 * (wno / 1e4) ** 4 for m in ("H2", "He")}
opa = px.RetrieveOpacities(wno, pt, molecular, continuum, cia_t, rayleigh_opa=ray, query_method="linear", ctx=ctx)
plev = np.logspace(-6, 2, nlevel)
prof = {"pressure": plev, "temperature": 150.0 + 1200.0 * ((np.log10(plev) + 6) / 8) ** 2, "H2": np.full(nlevel, 0.84),
        "He": np.full(nlevel, 0.155), "H2O": np.full(nlevel, 1e-3), "CH4": np.full(nlevel, 5e-4),
        "CO": np.full(nlevel, 1e-4), "NH3": np.full(nlevel, 1e-5)}
case = jdi.inputs()
case.phase_angle(0)
case.gravity(gravity=2500.0)
case.atmosphere(df=prof)
case.approx(raman="none")
if os.environ.get("STAR"):
    case.star(relative_flux=1.0 + 0.2 * np.cos(wno / 900.0), radius=6.9e10, semi_major=7.5e12)
    case.gravity(radius=7.1e9, mass=1.9e30)
for _ in range(30):
    case.spectrum(opa, calculation="reflected+thermal", devices=devs)
acc.clear()
tot = []
for _ in range(200):
    t0 = time.perf_counter()
    case.spectrum(opa, calculation="reflected+thermal", devices=devs)
    tot.append(time.perf_counter() - t0)
n = len(tot)
print("spectrum() median %.3f ms, min %.3f" % (1e3 * np.median(tot), 1e3 * min(tot)))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%-22s calls/spectrum %4.1f   ms/spectrum %.4f" % (k, len(v) / n, 1e3 * sum(v) / n))
