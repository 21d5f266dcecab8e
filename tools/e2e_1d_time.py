#!/usr/bin/env python
"""End-to-end time of the cloud-free 1-D spectrum(reflected+thermal) at 1e5 wavelengths x 90 layers (resident
synthetic opacity tables) -- run on the GPU box.  PROFILE_1D=1: many calls and nothing else (for rocprofv3
--kernel-trace --stats) plus a cProfile of the host side."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from picaso_amd import _lib
from picaso_amd import justdoit as jdi
from picaso_amd import optics as px

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
if os.environ.get("PHASE"):            # PHASE=<radians>: a 6 x 6 disk grid at that phase angle instead of the symmetric 1-D one
    case.phase_angle(float(os.environ["PHASE"]), num_gangle=6, num_tangle=6)
else:
    case.phase_angle(0)
case.gravity(gravity=2500.0)
case.atmosphere(df=prof)
# RAMAN=none|pollack|oklopcic, RT=toon|SH, LVL=1 (level fluxes), STAR=1 (a stellar spectrum on the grid: fpfs and Raman need one)
akw = dict(raman=os.environ.get("RAMAN", "none"), get_lvl_flux=bool(os.environ.get("LVL")))
if os.environ.get("RT", "toon") == "SH":
    akw.update(rt_method="SH", stream=4)
case.approx(**akw)
if os.environ.get("STAR") or akw["raman"] != "none" or "transmission" in os.environ.get("CALC", ""):
    case.star(relative_flux=1.0 + 0.2 * np.cos(wno / 900.0), radius=6.9e10, semi_major=7.5e12)
    case.gravity(radius=7.1e9, mass=1.9e30)
if akw["raman"] == "pollack":
    import tempfile
    d0 = tempfile.mkdtemp()
    os.makedirs(os.path.join(d0, "opacities"))
    wl = np.linspace(0.2, 6.0, 400)
    np.savetxt(os.path.join(d0, "opacities", "raman_fortran.txt"), np.column_stack([wl, 0.9 + 0.05 * np.cos(wl)]))
    os.environ["picaso_refdata"] = d0
if akw["raman"] == "oklopcic":
    g = np.load(os.path.join(ROOT, "tests", "golden", "optics.npz"))
    opa.raman_stellar_shifts = 1.0 + 0.02 * np.cos(np.outer(wno / 700.0, 1.0 + np.arange(len(g["in/raman_deltanu"]))))
    opa.raman_db = {"c": g["in/raman_c"], "ji": g["in/raman_ji"], "deltanu": g["in/raman_deltanu"]}
# CLOUD=box: a box cloud on a 196-point grid of its own (what virga and clouds(g0=..., p=..., dp=...) hand over),
# regridded per call -- on the device, or with PICASO_AMD_HOST_REGRID=1 by the reference's numpy.interp rows;
# CLOUD=table: the same cloud handed over as three (nlayer, nwno) host tables.
if os.environ.get("CLOUD"):
    import tempfile
    d = os.environ.get("picaso_refdata") or tempfile.mkdtemp()
    os.makedirs(os.path.join(d, "opacities"), exist_ok=True)
    wn = np.round(np.linspace(1900.0, 34000.0, 196)[::-1], 2)
    with open(os.path.join(d, "opacities", "wave_EGP.dat"), "w") as fh:
        fh.write("   i   micron.    wavenumber idum     idum1    idum2     idum3\n")
        for i, w in enumerate(wn):
            fh.write("%4d %9.3f %9.2f %8.2f- %7.2f %9.3f %9.3f\n" % (i + 1, 1e4 / w, w, w - 1, w + 1, 2.0, w))
    os.environ["picaso_refdata"] = d
    hk = dict(do_holes=True, fhole=0.3, fthin_cld=0.1) if os.environ.get("HOLES") else {}     # HOLES=1: patchy cloud
    case.clouds(g0=[0.8], w0=[0.95], opd=[1.5], p=[0.0], dp=[1.5], **hk)
    if os.environ["CLOUD"] == "table":
        cl = case.inputs["clouds"]
        case.clouds(df={k: np.stack([np.interp(wno, cl["wavenumber"], row) for row in cl["profile"][k]])
                        for k in ("opd", "w0", "g0")})
calc = os.environ.get("CALC", "reflected+thermal")
for _ in range(int(os.environ.get("WARM", "30"))):
    r = case.spectrum(opa, calculation=calc, full_output=bool(os.environ.get("FULL")))
if os.environ.get("PROFILE_1D"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    pdev = [int(x) for x in os.environ["DEVICES"].split(";")[0].split(",")] if os.environ.get("DEVICES") else None
    for _ in range(5):
        case.spectrum(opa, calculation=calc, devices=pdev)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        case.spectrum(opa, calculation=calc, full_output=bool(os.environ.get("FULL")), devices=pdev)
    pr.disable()
    pstats.Stats(pr).sort_stats(os.environ.get("SORT", "cumulative")).print_stats(int(os.environ.get("TOP", "35")))
    sys.exit(0)
out = {}
# DEVICES="0,0,0,0": the same spectrum in wavelength blocks (on one GPU: one context per entry), host cost of the cut
devsets = [None] + [[int(x) for x in d.split(",")] for d in os.environ.get("DEVICES", "").split(";") if d]
for devs in devsets:
    for _ in range(10):
        r = case.spectrum(opa, calculation=calc, devices=devs)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        r = case.spectrum(opa, calculation=calc, devices=devs, full_output=bool(os.environ.get("FULL")))
        ts.append(time.perf_counter() - t0)
    out["spectrum_1d_%d_%s_devices_%s_ms" % (nwno, calc, "none" if devs is None else len(devs))] = round(1e3 * min(ts), 3)
# BATCH="16,64": spectrum_batch() over that many copies of the case (each with its own temperature offset), BSIZE per
# launch (default 4), per spectrum
import copy
for B in [int(x) for x in os.environ.get("BATCH", "").split(",") if x]:
    cases = []
    for k in range(B):
        c = copy.deepcopy(case)
        pk = dict(prof, temperature=prof["temperature"] * (1.0 + 0.01 * k))
        c.atmosphere(df=pk)
        cases.append(c)
    for _ in range(3):
        rb = jdi.spectrum_batch(cases, opa, calculation=calc, batch_size=int(os.environ.get("BSIZE", "4")))
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        rb = jdi.spectrum_batch(cases, opa, calculation=calc, batch_size=int(os.environ.get("BSIZE", "4")))
        ts.append(time.perf_counter() - t0)
    tl = []
    for _ in range(3):
        t0 = time.perf_counter()
        rl = [c.spectrum(opa, calculation=calc) for c in cases]
        tl.append(time.perf_counter() - t0)
    same = all(np.array_equal(a[k], b[k]) for a, b in zip(rb, rl) for k in a if isinstance(a[k], np.ndarray))
    out["spectrum_batch_%d_ms_per_spectrum" % B] = round(1e3 * min(ts) / B, 3)
    out["spectrum_loop_%d_ms_per_spectrum" % B] = round(1e3 * min(tl) / B, 3)
    out["spectrum_batch_%d_equals_loop" % B] = bool(same)
out["albedo_sum"] = float(np.sum(r.get("albedo", 0.0)))
print(json.dumps(out))
