"""Seeded synthetic atmospheres for tests, golden fixtures and ``bench.py``.

No external data: the per-layer optical-depth components (gas, Rayleigh, cloud) are drawn from a
seeded generator and combined into the solver's input planes with the same algebra the reference
uses in ``compute_opacity`` (reference ``picaso/optics.py:329-354`` for the mixing and ``:401-420``
for the delta-Eddington scaling).  All planes are float64, C-order ``(nlayer|nlevel, nwno)`` --
layer-major, wavelength contiguous -- exactly what ``get_reflected_1d`` / ``get_thermal_1d``
receive from ``picaso()`` (reference ``picaso/justdoit.py:275-283, 337-342``).

This module is host-side input preparation only; it is not part of the accelerated path.
"""
import numpy as np

BASE_SEED = 20260928


def wavenumber_grid(nwno):
    """0.3-5 micron, linear in wavenumber (cm^-1)."""
    return np.linspace(2000.0, 33333.0, nwno)


def pressure_temperature(nlevel, kind="guillot"):
    """Level pressure (bar) and temperature (K), top -> bottom."""
    p = np.logspace(-6.0, 2.0, nlevel)
    x = (np.log10(p) + 6.0) / 8.0
    if kind == "jupiter":
        t = 110.0 + 55.0 * np.exp(-((x - 0.1) / 0.2) ** 2) + 900.0 * x ** 3.5
    else:  # monotone 150 K -> 1500 K
        t = 150.0 + 1350.0 * x ** 2.2
    return p, t


def tau_components(nlayer, nwno, seed, cloud=True, gas_scale=1.0, ray_scale=1.0,
                   cloud_opd=None):
    """Draw (TAUGAS, TAURAY, TAUCLD, w0_cld, g0_cld), each (nlayer, nwno)."""
    rng = np.random.default_rng(seed)
    nlevel = nlayer + 1
    wno = wavenumber_grid(nwno)
    p, _ = pressure_temperature(nlevel)
    dp = np.diff(p)                      # column density is proportional to dP
    # gas: separable a(layer)*b(wave) spanning ~1e-6..10^1.5 plus 0.3 dex noise
    a = np.log10(dp / dp.max())          # <= 0, deep layers ~0
    a = np.clip(a, -5.0, 0.0)
    b = 1.5 * np.sin(2.0 * np.pi * np.linspace(0.0, 6.0, nwno)) ** 2 - 1.5 \
        + 1.2 * np.cos(np.linspace(0.0, 40.0, nwno))
    lg = a[:, None] + b[None, :] + 0.3 * rng.standard_normal((nlayer, nwno)) + 1.0
    lg = np.clip(lg, -6.0, 1.5)
    taugas = gas_scale * 10.0 ** lg
    # Rayleigh ~ colden * wno^4, tau_Ray(0.4 um, 1 bar) ~ 0.2
    col = np.cumsum(dp)
    scale = 0.2 / (np.interp(1.0, p[1:], col) * (25000.0 / 1.0e4) ** 4)
    tauray = ray_scale * scale * dp[:, None] * (wno[None, :] / 1.0e4) ** 4
    taucld = np.zeros((nlayer, nwno))
    w0c = np.zeros((nlayer, nwno))
    g0c = np.zeros((nlayer, nwno))
    if cloud:
        nslab = max(1, min(10, nlayer // 3))
        top = int(0.55 * nlayer)
        sl = slice(top, min(nlayer, top + nslab))
        n = sl.stop - sl.start
        opd = cloud_opd if cloud_opd is not None else rng.uniform(0.01, 5.0)
        prof = rng.uniform(0.5, 1.5, size=(n, 1))
        spec = 1.0 + 0.3 * np.sin(np.linspace(0.0, 9.0, nwno))[None, :]
        taucld[sl] = opd / n * prof * spec
        w0c[sl] = rng.uniform(0.5, 0.999, size=(n, 1)) * (1.0 - 0.05 * rng.random((n, nwno)))
        g0c[sl] = rng.uniform(0.0, 0.9, size=(n, 1)) * (1.0 - 0.05 * rng.random((n, nwno)))
    return taugas, tauray, taucld, w0c, g0c


def mix_planes(taugas, tauray, taucld, w0c, g0c, raman_factor=0.99999, delta_eddington=True,
               stream=2):
    """Combine tau components into the solver planes.

    Follows reference ``optics.py:329-354`` (DTAU, ftau_cld, ftau_ray, GCOS2, W0, W0_no_raman, TAU)
    and ``:401-420`` (delta-Eddington).  Returns a dict with the 13 arrays ``compute_opacity``
    returns (ngauss axis dropped).
    """
    nlayer, nwno = taugas.shape
    dtau_og = taugas + tauray + taucld
    with np.errstate(invalid="ignore", divide="ignore"):
        ftau_cld = (w0c * taucld) / (w0c * taucld + tauray)
        ftau_ray = tauray / (tauray + w0c * taucld)
    gcos2 = 0.5 * ftau_ray
    cosb_og = g0c.copy()
    w0_og = (tauray * raman_factor + taucld * w0c) / dtau_og
    w0_no_raman = (tauray * 0.99999 + taucld * w0c) / dtau_og
    tau_og = np.zeros((nlayer + 1, nwno))
    tau_og[1:] = np.cumsum(dtau_og, axis=0)
    if delta_eddington:
        f = cosb_og ** stream
        w0 = w0_og * (1.0 - f) / (1.0 - w0_og * f)
        cosb = (cosb_og - f) / (1.0 - f)
        dtau = dtau_og * (1.0 - w0_og * f)
        tau = np.zeros((nlayer + 1, nwno))
        tau[1:] = np.cumsum(dtau, axis=0)
        f_deltaM = f
    else:
        w0, cosb, dtau, tau = w0_og, cosb_og, dtau_og, tau_og
        f_deltaM = 0.0 * cosb_og
    out = dict(dtau=dtau, tau=tau, w0=w0, cosb=cosb, ftau_cld=ftau_cld, ftau_ray=ftau_ray,
               gcos2=gcos2, dtau_og=dtau_og, tau_og=tau_og, w0_og=w0_og, cosb_og=cosb_og,
               w0_no_raman=w0_no_raman, f_deltaM=f_deltaM)
    return {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in out.items()}


def make_scene(nlayer, nwno, seed=0, cloud=True, delta_eddington=True, stream=2,
               gas_scale=1.0, ray_scale=1.0, cloud_opd=None, tkind="guillot"):
    """Full synthetic scene: planes + grid + T/P profile."""
    comps = tau_components(nlayer, nwno, BASE_SEED + seed, cloud=cloud, gas_scale=gas_scale,
                           ray_scale=ray_scale, cloud_opd=cloud_opd)
    planes = mix_planes(*comps, delta_eddington=delta_eddington, stream=stream)
    p, t = pressure_temperature(nlayer + 1, tkind)
    planes.update(wno=wavenumber_grid(nwno), plevel=p * 1.0e6, tlevel=t,
                  nlayer=nlayer, nlevel=nlayer + 1, nwno=nwno)
    # keep the un-mixed components so a device-side compute_opacity can be checked later
    planes.update(taugas=comps[0], tauray=comps[1], taucld=comps[2], w0_cld=comps[3],
                  g0_cld=comps[4])
    return planes


def constant_scene(nlayer, nwno, dtau, w0, g0, rayleigh=False):
    """``test_mode`` style scene (reference ``optics.py:372-399``): constant dtau/w0/g0 planes."""
    shp = (nlayer, nwno)
    d = np.full(shp, float(dtau)) if np.isscalar(dtau) else np.repeat(
        np.asarray(dtau, float)[:, None], nwno, axis=1)
    d = np.where(d <= 0, 1e-10, d)
    w = np.full(shp, max(float(w0), 1e-10))
    g = np.full(shp, float(g0))
    if rayleigh:
        gcos2, fr, fc = np.full(shp, 0.5), np.ones(shp), np.zeros(shp)
    else:
        gcos2, fr, fc = np.zeros(shp), np.zeros(shp), np.ones(shp)
    tau = np.zeros((nlayer + 1, nwno))
    tau[1:] = np.cumsum(d, axis=0)
    return dict(dtau_og=d, tau_og=tau, w0_og=w, cosb_og=g, ftau_cld=fc, ftau_ray=fr, gcos2=gcos2,
                w0_no_raman=w.copy())


def delta_scale(sc, delta_eddington=True, stream=2):
    """Apply reference ``optics.py:401-431`` to a constant_scene dict (adds dtau,tau,w0,cosb,f_deltaM)."""
    out = dict(sc)
    if delta_eddington:
        f = sc["cosb_og"] ** stream
        out["w0"] = sc["w0_og"] * (1.0 - f) / (1.0 - sc["w0_og"] * f)
        out["cosb"] = (sc["cosb_og"] - f) / (1.0 - f)
        out["dtau"] = sc["dtau_og"] * (1.0 - sc["w0_og"] * f)
        tau = np.zeros_like(sc["tau_og"])
        tau[1:] = np.cumsum(out["dtau"], axis=0)
        out["tau"] = tau
        out["f_deltaM"] = f
    else:
        out.update(w0=sc["w0_og"].copy(), cosb=sc["cosb_og"].copy(), dtau=sc["dtau_og"].copy(),
                   tau=sc["tau_og"].copy(), f_deltaM=0.0 * sc["cosb_og"])
    return out


def opacity_tables(nwno, mols=("H2O", "CH4"), wno=None):
    """Arguments of ``optics.RetrieveOpacities(...)`` for a smooth synthetic monochromatic database on ``nwno``
    wavelengths: a 5 x 6 (T, P) grid of molecular cross sections, two CIA pairs on 6 temperatures, Rayleigh
    cross sections of H2 / He / CH4 -- the inputs of the 3-D / end-to-end workloads (BASELINE configs[4]: 64
    facet plane sets generated on the device from per-facet profiles)."""
    wno = np.linspace(3000.0, 30000.0, nwno) if wno is None else np.asarray(wno, dtype=float)
    temps, press = [100.0, 300.0, 700.0, 1500.0, 3000.0], [1e-6, 1e-4, 1e-2, 1.0, 100.0, 500.0]
    pt, molecular, pid = [], {m: {} for m in mols}, 0
    for t in temps:
        for p in press:
            pid += 1
            pt.append((pid, p, t))
            for i, m in enumerate(mols):
                molecular[m][pid] = 10.0 ** (-24 + 2 * np.sin(wno / 2500.0 + i) + 0.4 * np.log10(p)
                                             + 0.8 * np.log10(t / 300.0))
    cia_t = [75.0, 200.0, 500.0, 1000.0, 2000.0, 4000.0]
    continuum = {pr: {t: 10.0 ** (-7 + np.cos(wno / 4000.0 + j) + 0.3 * np.log10(t / 300.0)) for t in cia_t}
                 for j, pr in enumerate(("H2H2", "H2He"))}
    ray = {m: 1e-27 * (wno / 1e4) ** 4 * (1 + 0.1 * k) for k, m in enumerate(("H2", "He", "CH4"))}
    return dict(wno=wno, pt_pairs=pt, molecular=molecular, continuum=continuum, cia_temps=cia_t, rayleigh_opa=ray)


def facet_profiles(nlevel, ng, nt, mols=("H2O", "CH4"), amplitude=0.1):
    """Level profile with per-facet temperatures perturbed by ``amplitude`` (SURVEY 8(d), configs[4])."""
    plev = np.logspace(-6, 2, nlevel)
    t = 150.0 + 1200.0 * ((np.log10(plev) + 6) / 8) ** 2
    pert = 1.0 + amplitude * np.cos(np.arange(ng * nt).reshape(ng, nt))
    prof = {"pressure": plev, "temperature": t[:, None, None] * pert[None], "H2": np.full(nlevel, 0.84),
            "He": np.full(nlevel, 0.155)}
    for i, m in enumerate(mols):
        prof[m] = np.full(nlevel, 1e-3 / (2 ** i))
    return prof


def cloud_slab(nlayer, nwno, opd=0.4, w0=0.95, g0=0.7, top=0.55, thickness=10):
    """One ``(nlayer, nwno)`` cloud table (the same on every facet): a slab of ``thickness`` layers."""
    a = int(top * nlayer)
    sl = slice(a, min(nlayer, a + thickness))
    out = {k: np.zeros((nlayer, nwno)) for k in ("opd", "w0", "g0")}
    ramp = np.linspace(0.8, 1.2, nwno)[None, :]
    out["opd"][sl] = opd * ramp
    out["w0"][sl] = w0
    out["g0"][sl] = g0
    return out
