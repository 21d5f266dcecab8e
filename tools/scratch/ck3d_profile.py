#!/usr/bin/env python
"""cProfile of spectrum(dimension='3d') and of an 8-phase thermal curve on correlated-k tables (661 x 8, 64 facets):
bench.py's product.correlated_k workload (run on the GPU box)."""
import cProfile, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from picaso_amd import _lib
from picaso_amd import justdoit as jdi
from picaso_amd import optics as px
ctx = _lib.context(0)
nlevel = 91
temps, press = [100.0, 300.0, 700.0, 1500.0, 3000.0], [1e-6, 1e-4, 1e-2, 1e-1, 1.0, 10.0, 100.0, 500.0]
cia_t = [75.0, 200.0, 500.0, 1000.0, 2000.0, 4000.0]
plev = np.logspace(-6, 2, nlevel)
prof = {"pressure": plev, "temperature": 150.0 + 1200.0 * ((np.log10(plev) + 6) / 8) ** 2,
        "H2": np.full(nlevel, 0.84), "He": np.full(nlevel, 0.155), "H2O": np.full(nlevel, 1e-3),
        "CH4": np.full(nlevel, 5e-4), "CO": np.full(nlevel, 1e-4), "NH3": np.full(nlevel, 1e-5)}
nb, nk = 661, 8
wck = np.linspace(40.0, 28000.0, nb)
xg, wg = np.polynomial.legendre.leggauss(4)
gpts = np.concatenate([0.95 * 0.5 * (xg + 1), 0.95 + 0.05 * 0.5 * (xg + 1)])
gwts = np.concatenate([0.95 * 0.5 * wg, 0.05 * 0.5 * wg])
tk, pk = np.array(temps), np.array(press)
lnk = np.log(10.0) * (-26.0 + 2.0 * np.sin(wck / 2500.0)[None, None, :, None] + 0.5 * np.log10(pk)[:, None, None, None]
                      + 0.9 * np.log10(tk / 300.0)[None, :, None, None] + 0.6 * np.arange(nk)[None, None, None, :])
cont = {pr: {t: 10.0 ** (-7.0 + np.cos(wck / 4000.0 + k) + 0.3 * np.log10(t / 300.0)) for t in cia_t}
        for k, pr in enumerate(("H2H2", "H2He"))}
opk = px.RetrieveCKs(wck, gwts, np.tile(pk, tk.size), np.repeat(tk, pk.size), np.full(tk.size, pk.size), lnk,
                     continuum=cont, cia_temps=cia_t, rayleigh_opa={m: 1e-27 * (wck / 1e4) ** 4 for m in ("H2", "He")},
                     gauss_pts=gpts, ctx=ctx)
pert = 1.0 + 0.1 * np.cos(np.arange(64).reshape(8, 8))
c3 = jdi.inputs()
c3.phase_angle(np.pi / 3, num_gangle=8, num_tangle=8)
c3.gravity(gravity=2500.0)
c3.atmosphere_3d(dict(prof, temperature=prof["temperature"][:, None, None] * pert[None]))
c3.approx(raman="none")
calc = os.environ.get("CALC", "reflected+thermal")
if os.environ.get("DIM") == "1d":
    c1 = jdi.inputs()
    c1.phase_angle(0, num_gangle=5)
    c1.gravity(gravity=2500.0)
    c1.atmosphere(df=dict(prof))
    c1.approx(raman="none")
    for _ in range(5):
        c1.spectrum(opk, calculation=calc)
    tt = []
    for _ in range(20):
        t0 = time.perf_counter(); c1.spectrum(opk, calculation=calc); tt.append(time.perf_counter() - t0)
    print("1-D CK spectrum ms", 1e3 * np.median(tt))
    sys.exit(0)
for _ in range(3):
    c3.spectrum(opk, calculation=calc, dimension="3d")
tt = []
for _ in range(8):
    t0 = time.perf_counter(); c3.spectrum(opk, calculation=calc, dimension="3d"); tt.append(time.perf_counter() - t0)
print("3-D CK spectrum ms", 1e3 * np.median(tt))
if os.environ.get("PROFILE", "1") == "1":
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5):
        c3.spectrum(opk, calculation=calc, dimension="3d")
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
if os.environ.get("PHASES"):
    phases = list(2 * np.pi * (np.arange(8) + 0.5) / 8)
    pc = jdi.inputs()
    pc.phase_curve_geometry("thermal", phases, num_gangle=8, num_tangle=8)
    pc.gravity(gravity=2500.0)
    pc.atmosphere_4d([dict(prof, temperature=prof["temperature"][:, None, None] * (pert[None] + 0.01 * k)) for k in range(8)])
    pc.approx(raman="none")
    pc.phase_curve(opk)
    t0 = time.perf_counter(); pc.phase_curve(opk); print("thermal CK phase curve ms", 1e3 * (time.perf_counter() - t0))
    pr = cProfile.Profile(); pr.enable(); pc.phase_curve(opk); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
