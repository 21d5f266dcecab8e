#!/usr/bin/env python
"""End-to-end time of spectrum(dimension='3d') at the BASELINE configs[4] per-GPU shard (8 x 8 facets x
12 500 wavelengths x 90 layers, per-facet temperatures, reflected + thermal) and of a phase curve --
run on the GPU box.  PICASO_AMD_FACET_LOOP=1 gives the per-facet-loop path for comparison."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from picaso_amd import _lib
from picaso_amd import justdoit as jdi
from picaso_amd import optics as px

nwno, nlevel = int(os.environ.get("NWNO", "12500")), 91
ctx = _lib.context(0)
rng = np.random.default_rng(1)
wno = np.linspace(3000.0, 30000.0, nwno)
temps, press = [100.0, 300.0, 700.0, 1500.0, 3000.0], [1e-6, 1e-4, 1e-2, 1.0, 100.0, 500.0]
mols = ["H2O", "CH4", "CO", "NH3", "H2"]
pt, molecular, pid = [], {m: {} for m in mols}, 0
for t in temps:
    for p in press:
        pid += 1
        pt.append((pid, p, t))
        for i, m in enumerate(mols):
            molecular[m][pid] = 10.0 ** (-24 + 2 * np.sin(wno / 2500.0 + i) + 0.4 * np.log10(p) + 0.8 * np.log10(t / 300.0))
cia_t = [75.0, 200.0, 500.0, 1000.0, 2000.0, 4000.0]
continuum = {pr: {t: 10.0 ** (-7 + np.cos(wno / 4000.0 + j) + 0.3 * np.log10(t / 300.0)) for t in cia_t}
             for j, pr in enumerate(("H2H2", "H2He"))}
ray = {m: 1e-27 * (wno / 1e4) ** 4 * (1 + 0.1 * k) for k, m in enumerate(("H2", "He", "CH4"))}
opa = px.RetrieveOpacities(wno, pt, molecular, continuum, cia_t, rayleigh_opa=ray, query_method="linear", ctx=ctx)
plev = np.logspace(-6, 2, nlevel)
prof = {"pressure": plev, "temperature": 150.0 + 1200.0 * ((np.log10(plev) + 6) / 8) ** 2, "H2": np.full(nlevel, 0.84),
        "He": np.full(nlevel, 0.155), "H2O": np.full(nlevel, 1e-3), "CH4": np.full(nlevel, 5e-4),
        "CO": np.full(nlevel, 1e-4), "NH3": np.full(nlevel, 1e-5)}
ng = nt = 8
pert = 1.0 + 0.1 * np.cos(np.arange(64).reshape(8, 8))
prof3 = dict(prof)
prof3["temperature"] = prof["temperature"][:, None, None] * pert[None]
c3 = jdi.inputs()
c3.phase_angle(np.pi / 3, num_gangle=ng, num_tangle=nt)
c3.gravity(gravity=2500.0)
c3.atmosphere_3d(prof3)
c3.approx(raman="none")
if os.environ.get("CLOUD3D"):         # CLOUD3D=1: a grey cloud below layer 50 on every facet, tables on a 196-point grid of their own
    box = np.zeros((nlevel - 1, 196))
    box[50:60] = 0.3
    if os.environ.get("CLOUD3D") == "2":   # ... differing from facet to facet: (nlayer, 196, ng, nt) tables
        box = box[:, :, None, None] * (1.0 + 0.3 * np.cos(np.arange(64).reshape(1, 1, 8, 8)))
    c3.clouds_3d(df={"opd": box, "w0": np.where(box > 0, 0.95, 0.0), "g0": np.where(box > 0, 0.6, 0.0),
                     "wavenumber": np.linspace(wno[0], wno[-1], 196)})
if os.environ.get("PROFILE_3D"):      # PROFILE_3D=1: cProfile + wall time of the batched call alone (for rocprofv3 too)
    import cProfile, pstats
    for _ in range(30):
        c3.spectrum(opa, calculation="reflected+thermal", dimension="3d")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        c3.spectrum(opa, calculation="reflected+thermal", dimension="3d")
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
    sys.exit(0)
out = {}
for tag, env in (("batched", None), ("per_facet_loop", "1")):
    if env:
        os.environ["PICASO_AMD_FACET_LOOP"] = env
    else:
        os.environ.pop("PICASO_AMD_FACET_LOOP", None)
    r = c3.spectrum(opa, calculation="reflected+thermal", dimension="3d")
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        r = c3.spectrum(opa, calculation="reflected+thermal", dimension="3d")
        ts.append(time.perf_counter() - t0)
    out["spectrum_3d_8x8x%d_%s_ms" % (nwno, tag)] = round(1e3 * min(ts), 3)
    out["albedo_sum_" + tag] = float(np.sum(r["albedo"]))
os.environ.pop("PICASO_AMD_FACET_LOOP", None)
for calc in ("reflected", "thermal"):
    c3.spectrum(opa, calculation=calc, dimension="3d")
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        c3.spectrum(opa, calculation=calc, dimension="3d")
        ts.append(time.perf_counter() - t0)
    out["spectrum_3d_%s_only_ms" % calc] = round(1e3 * min(ts), 3)
if os.environ.get("CLOUD3D"):         # ... and an 8-phase reflected curve with that cloud map at every phase
    P = 8
    phases = list(np.linspace(0.0, 2 * np.pi * (P - 1) / P, P))
    pc = jdi.inputs()
    pc.phase_curve_geometry("reflected", phases, num_gangle=ng, num_tangle=nt)
    pc.gravity(gravity=2500.0)
    pc.atmosphere_4d([prof3 for _ in phases])
    pc.approx(raman="none")
    cmap = c3.inputs["clouds"]["profile_3d"]
    pc.phase_curve(opa, clouds_by_phase=[cmap] * P)
    if os.environ.get("PROFILE_PC"):  # PROFILE_PC=1: cProfile of the cloudy curve
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
        pc.phase_curve(opa, clouds_by_phase=[cmap] * P)
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
    t0 = time.perf_counter()
    res = pc.phase_curve(opa, clouds_by_phase=[cmap] * P)
    out["phase_curve_reflected_%d_phases_cloudy_ms" % P] = round(1e3 * (time.perf_counter() - t0), 3)
    print(json.dumps(out))
    sys.exit(0)
# phase curve: P phases of the same map
P = 8
phases = list(np.linspace(0.0, 2 * np.pi * (P - 1) / P, P))
pc = jdi.inputs()
pc.phase_curve_geometry("reflected", phases, num_gangle=ng, num_tangle=nt)
pc.gravity(gravity=2500.0)
pc.atmosphere_4d([prof3 for _ in phases])
pc.approx(raman="none")
pc.phase_curve(opa)
t0 = time.perf_counter()
res = pc.phase_curve(opa)
out["phase_curve_reflected_%d_phases_ms" % P] = round(1e3 * (time.perf_counter() - t0), 3)
pt = jdi.inputs()
pt.phase_curve_geometry("thermal", phases, num_gangle=ng, num_tangle=nt)
pt.gravity(gravity=2500.0)
pt.atmosphere_4d([prof3 for _ in phases])
pt.approx(raman="none")
pt.phase_curve(opa)
t0 = time.perf_counter()
res = pt.phase_curve(opa)
out["phase_curve_thermal_%d_phases_ms" % P] = round(1e3 * (time.perf_counter() - t0), 3)
print(json.dumps(out))
