"""Drop-in counterparts of the reference's radiative-transfer functions (``picaso/fluxes.py``).

Same names, positional argument order, dtypes/shapes and return tuples as the reference; the
arithmetic runs in the hand-written gfx950 kernels behind ``include/picaso_hip.h``.  Inputs may
be numpy arrays (copied to HBM, result copied back) -- exactly how ``justdoit.picaso()`` calls the
reference (reference picaso/justdoit.py:275-283, 337-342, 492, 510).  For HBM-resident planes use
``picaso_amd.resident``.
"""
import ctypes

import numpy as np

from ._lib import check, context, f64, load, per_wave, ptr, serialized

_ci, _cd = ctypes.c_int, ctypes.c_double


@serialized
def get_reflected_1d(nlevel, wno, nwno, numg, numt, dtau, tau, w0, cosb, gcos2, ftau_cld, ftau_ray,
                     dtau_og, tau_og, w0_og, cosb_og, surf_reflect, ubar0, ubar1, cos_theta, F0PI,
                     single_phase, multi_phase, frac_a, frac_b, frac_c, constant_back,
                     constant_forward, get_toa_intensity=1, get_lvl_flux=0, toon_coefficients=0,
                     b_top=0):
    """Toon89 reflected light, 1-D (reference ``fluxes.get_reflected_1d``, fluxes.py:1009-1413).

    Returns ``(xint_at_top (numg,numt,nwno), (flux_minus_all, flux_plus_all, flux_minus_midpt_all,
    flux_plus_midpt_all))`` with the four ``(numg,numt,nlevel,nwno)`` arrays zero unless
    ``get_lvl_flux`` (as in the reference).
    """
    ctx = context()
    nlayer = nlevel - 1
    planes = [f64(p) for p in (dtau, tau, w0, cosb, gcos2, ftau_cld, ftau_ray, dtau_og, tau_og,
                               w0_og, cosb_og)]
    for p, rows in zip(planes, (nlayer, nlevel, nlayer, nlayer, nlayer, nlayer, nlayer, nlayer,
                                nlevel, nlayer, nlayer)):
        if p.shape != (rows, nwno):
            raise Exception("get_reflected_1d: plane of shape %s, expected %s" % (p.shape, (rows, nwno)))
    rs, f0 = per_wave(surf_reflect, nwno), per_wave(F0PI, nwno)
    u0, u1 = f64(ubar0, (numg, numt)), f64(ubar1, (numg, numt))
    xint = np.zeros((numg, numt, nwno))
    lvl = [np.zeros((numg, numt, nlevel, nwno)) for _ in range(4)]
    check(load().picaso_get_reflected_1d(
        ctx, _ci(nlevel), None, _ci(nwno), _ci(numg), _ci(numt), *[ptr(p) for p in planes],
        ptr(rs), ptr(u0), ptr(u1), _cd(cos_theta), ptr(f0), _ci(int(single_phase)),
        _ci(int(multi_phase)), _cd(frac_a), _cd(frac_b), _cd(frac_c), _cd(constant_back),
        _cd(constant_forward), _ci(int(get_toa_intensity)), _ci(int(get_lvl_flux)),
        _ci(int(toon_coefficients)), _cd(b_top), ptr(xint),
        *[ptr(l) if get_lvl_flux else None for l in lvl]), ctx)
    return xint, tuple(lvl)


@serialized
def get_reflected_3d(nlevel, wno, nwno, numg, numt, dtau_3d, tau_3d, w0_3d, cosb_3d, gcos2_3d,
                     ftau_cld_3d, ftau_ray_3d, dtau_og_3d, tau_og_3d, w0_og_3d, cosb_og_3d,
                     surf_reflect, ubar0, ubar1, cos_theta, F0PI, single_phase, multi_phase, frac_a,
                     frac_b, frac_c, constant_back, constant_forward):
    """Toon89 reflected light with per-facet planes ``(nlayer|nlevel, nwno, numg, numt)``
    (reference ``fluxes.get_reflected_3d``, fluxes.py:354-660).  Returns ``xint_at_top``."""
    ctx = context()
    planes = [f64(p) for p in (dtau_3d, tau_3d, w0_3d, cosb_3d, gcos2_3d, ftau_cld_3d, ftau_ray_3d,
                               dtau_og_3d, tau_og_3d, w0_og_3d, cosb_og_3d)]
    rs, f0 = per_wave(surf_reflect, nwno), per_wave(F0PI, nwno)
    u0, u1 = f64(ubar0, (numg, numt)), f64(ubar1, (numg, numt))
    xint = np.zeros((numg, numt, nwno))
    check(load().picaso_get_reflected_3d(
        ctx, _ci(nlevel), None, _ci(nwno), _ci(numg), _ci(numt), *[ptr(p) for p in planes],
        ptr(rs), ptr(u0), ptr(u1), _cd(cos_theta), ptr(f0), _ci(int(single_phase)),
        _ci(int(multi_phase)), _cd(frac_a), _cd(frac_b), _cd(frac_c), _cd(constant_back),
        _cd(constant_forward), ptr(xint)), ctx)
    return xint


@serialized
def get_thermal_1d(nlevel, wno, nwno, numg, numt, tlevel, dtau, w0, cosb, plevel, ubar1,
                   surf_reflect, hard_surface, dwno, calc_type, want_lvl=True):
    """Toon89 thermal emission, 1-D (reference ``fluxes.get_thermal_1d``, fluxes.py:1682-1912).

    Returns ``(flux_at_top, (flux_minus, flux_plus, flux_minus_mdpt, flux_plus_mdpt))``.  The
    reference always fills the four ``(numg,numt,nlevel,nwno)`` arrays; ``want_lvl=False`` (what
    this package's own ``picaso()`` passes for a spectrum) skips them and returns zeros there.
    """
    ctx = context()
    wno_, tl, pl = f64(wno), f64(tlevel), f64(plevel)
    dt, w0_, cb = f64(dtau), f64(w0), f64(cosb)
    rs, dw = per_wave(surf_reflect, nwno), per_wave(dwno, nwno)
    u1 = f64(ubar1, (numg, numt))
    flux = np.zeros((numg, numt, nwno))
    lvl = [np.zeros((numg, numt, nlevel, nwno)) for _ in range(4)]
    check(load().picaso_get_thermal_1d(
        ctx, _ci(nlevel), ptr(wno_), _ci(nwno), _ci(numg), _ci(numt), ptr(tl), ptr(dt), ptr(w0_),
        ptr(cb), ptr(pl), ptr(u1), ptr(rs), _ci(int(hard_surface)), ptr(dw), _ci(int(calc_type)),
        ptr(flux), *[ptr(l) if want_lvl else None for l in lvl]), ctx)
    return flux, tuple(lvl)


@serialized
def get_thermal_3d(nlevel, wno, nwno, numg, numt, tlevel_3d, dtau_3d, w0_3d, cosb_3d, plevel_3d,
                   ubar1, surf_reflect, hard_surface):
    """Toon89 thermal emission with per-facet profiles (reference ``fluxes.get_thermal_3d``,
    fluxes.py:2147-2352).  Returns ``int_at_top (numg,numt,nwno)``."""
    ctx = context()
    wno_ = f64(wno)
    tl, pl = f64(tlevel_3d), f64(plevel_3d)
    dt, w0_, cb = f64(dtau_3d), f64(w0_3d), f64(cosb_3d)
    rs = per_wave(surf_reflect, nwno)
    u1 = f64(ubar1, (numg, numt))
    out = np.zeros((numg, numt, nwno))
    check(load().picaso_get_thermal_3d(
        ctx, _ci(nlevel), ptr(wno_), _ci(nwno), _ci(numg), _ci(numt), ptr(tl), ptr(dt), ptr(w0_),
        ptr(cb), ptr(pl), ptr(u1), ptr(rs), _ci(int(hard_surface)), ptr(out)), ctx)
    return out


@serialized
def get_reflected_SH(nlevel, nwno, numg, numt, dtau, tau, w0, cosb, ftau_cld, ftau_ray, f_deltaM,
                     dtau_og, tau_og, w0_og, cosb_og, surf_reflect, ubar0, ubar1, cos_theta, F0PI,
                     w_single_form, w_multi_form, psingle_form, w_single_rayleigh, w_multi_rayleigh,
                     psingle_rayleigh, frac_a, frac_b, frac_c, constant_back, constant_forward,
                     stream, b_top=0, flx=0, single_form=0, compound_f_deltaM=True):
    """Spherical-harmonics reflected light, stream = 2 or 4 (reference ``fluxes.get_reflected_SH``,
    fluxes.py:2675-2976).  Returns ``(xint_at_top, flux)``; ``flux`` is
    ``(numg, numt, stream*nlevel, nwno)``, zeros for ``flx=0`` as in the reference and the layer moment
    fluxes ``F.X + G`` of ``calculate_flux`` (fluxes.py:3631-3635) for ``flx=1``.

    ``compound_f_deltaM=True`` (default) reproduces the reference: its TTHG branch multiplies
    ``f_deltaM`` in place once per angle (fluxes.py:2823-2824), so angle k sees
    ``f_deltaM * fac**(k+1)`` and the caller's array is left multiplied by ``fac**(numg*numt)``
    (done here too when ``f_deltaM`` is a writable float64 array).  ``False`` gives the
    non-compounding variant (every angle sees ``f_deltaM * fac``; the caller's array is untouched).

    Not in the reference: a cloud-free atmosphere may pass ``None`` for every plane but ``dtau`` and ``w0`` (stream 4,
    default options, ``flx=0``; ``picaso_reflected_SH_can_derive``) -- the launch then shares the angle-independent
    half of each layer between the disk angles.
    """
    ctx = context()
    arrs = [f64(p) if p is not None else None
            for p in (dtau, tau, w0, cosb, ftau_cld, ftau_ray, f_deltaM, dtau_og, tau_og, w0_og, cosb_og)]
    rs, f0 = per_wave(surf_reflect, nwno), per_wave(F0PI, nwno)
    u0, u1 = f64(ubar0, (numg, numt)), f64(ubar1, (numg, numt))
    xint = np.zeros((numg, numt, nwno))
    flux = np.zeros((numg, numt, stream * nlevel, nwno))
    check(load().picaso_get_reflected_SH(
        ctx, _ci(nlevel), _ci(nwno), _ci(numg), _ci(numt), *[ptr(p) for p in arrs], ptr(rs), ptr(u0),
        ptr(u1), _cd(cos_theta), ptr(f0), _ci(int(w_single_form)), _ci(int(w_multi_form)),
        _ci(int(psingle_form)), _ci(int(w_single_rayleigh)), _ci(int(w_multi_rayleigh)),
        _ci(int(psingle_rayleigh)), _cd(frac_a), _cd(frac_b), _cd(frac_c), _cd(constant_back),
        _cd(constant_forward), _ci(int(stream)), _cd(b_top), _ci(int(flx)), _ci(int(single_form)),
        _ci(1 if compound_f_deltaM else 0), ptr(xint), ptr(flux) if flx else None), ctx)
    if compound_f_deltaM and (w_single_form == 0 or w_multi_form == 0) and \
            isinstance(f_deltaM, np.ndarray) and f_deltaM.dtype == np.float64 and f_deltaM.flags.writeable:
        gb = constant_back * np.asarray(cosb_og, dtype=float)
        f = frac_a + frac_b * gb ** frac_c
        f_deltaM *= (f * constant_forward ** stream + (1 - f) * constant_back ** stream) ** (numg * numt)
    return xint, flux


@serialized
def get_thermal_SH(nlevel, wno, nwno, numg, numt, tlevel, dtau, tau, w0, cosb, dtau_og, tau_og,
                   w0_og, w0_no_raman, cosb_og, plevel, ubar1, surf_reflect, stream, hard_surface,
                   flx=0):
    """Spherical-harmonics thermal emission (reference ``fluxes.get_thermal_SH``,
    fluxes.py:2979-3186; ``flx=1`` is broken in the reference and raises here).
    Returns ``(xint_at_top, flux)``."""
    ctx = context()
    wno_, tl, pl = f64(wno), f64(tlevel), f64(plevel)
    dt, ta, w0_, cbo = f64(dtau), f64(tau), f64(w0), f64(cosb_og)
    differs = 0 if np.array_equal(np.asarray(cosb), np.asarray(cosb_og)) else 1   # fluxes.py:3072
    rs = per_wave(surf_reflect, nwno)
    u1 = f64(ubar1, (numg, numt))
    xint = np.zeros((numg, numt, nwno))
    check(load().picaso_get_thermal_SH(
        ctx, _ci(nlevel), ptr(wno_), _ci(nwno), _ci(numg), _ci(numt), ptr(tl), ptr(dt), ptr(ta),
        ptr(w0_), ptr(cbo), ptr(pl), ptr(u1), ptr(rs), _ci(int(stream)), _ci(int(hard_surface)),
        _ci(differs), _ci(int(flx)), ptr(xint)), ctx)
    return xint, np.zeros((numg, numt, stream * nlevel, nwno))


@serialized
def get_transit_1d(z, dz, nlevel, nwno, rstar, mmw, k_b, amu, player, tlayer, colden, DTAU):
    """Transmission spectrum ``(Rp/Rs)**2`` (reference ``fluxes.get_transit_1d``, fluxes.py:2581-2663).
    Same positional arguments; ``DTAU`` is ``(nlayer, nwno)``.  Returns ``(nwno,)``."""
    ctx = context()
    d = f64(DTAU)
    if d.shape != (nlevel - 1, nwno):
        raise Exception("get_transit_1d: DTAU of shape %s, expected %s" % (d.shape, (nlevel - 1, nwno)))
    out = np.zeros(nwno)
    check(load().picaso_get_transit_1d(
        ctx, ptr(f64(z, (nlevel,))), ptr(f64(dz, (nlevel,))), _ci(nlevel), _ci(nwno), _cd(rstar),
        ptr(f64(mmw, (nlevel - 1,))), _cd(k_b), _cd(amu), ptr(f64(player)), ptr(f64(tlayer)),
        ptr(f64(colden, (nlevel - 1,))), ptr(d), ptr(out)), ctx)
    return out


@serialized
def blackbody(t, w):
    """Planck function per unit wavelength in cgs (reference ``fluxes.blackbody``, fluxes.py:1660-1680): ``t`` in K,
    ``w`` WAVELENGTH in cm; returns ``(ntemp, nwave)``.  The same device function the thermal solvers evaluate level by
    level (``get_thermal_1d`` forms ``blackbody(tlevel, 1/wno)``, fluxes.py:1752)."""
    ctx = context()
    t_ = np.ascontiguousarray(np.atleast_1d(t), dtype=np.float64).ravel()
    w_ = np.ascontiguousarray(np.atleast_1d(w), dtype=np.float64).ravel()
    out = np.zeros((t_.size, w_.size))
    check(load().picaso_blackbody(ctx, _ci(t_.size), ptr(t_), ctypes.c_long(w_.size), ptr(w_), ptr(out)), ctx)
    return out


@serialized
def blackbody_integrated(T, wave, dwave):
    """Mean of the wavenumber Planck function over each bin (three points: centre and both edges), the climate
    calculation's blackbody (reference ``fluxes.blackbody_integrated``, fluxes.py:1609-1658): ``T`` in K, ``wave`` and
    ``dwave`` in cm^-1; returns ``(ntemp, nwave)``."""
    ctx = context()
    t_ = np.ascontiguousarray(np.atleast_1d(T), dtype=np.float64).ravel()
    w_ = np.ascontiguousarray(np.atleast_1d(wave), dtype=np.float64).ravel()
    d_ = f64(np.atleast_1d(dwave), (w_.size,))
    out = np.zeros((t_.size, w_.size))
    check(load().picaso_blackbody_integrated(ctx, _ci(t_.size), ptr(t_), ctypes.c_long(w_.size), ptr(w_), ptr(d_),
                                             ptr(out)), ctx)
    return out


def chapman(pressure, pm, hratio):
    """Chapman function of the tidal / energy-injection profile (reference ``fluxes.chapman``, fluxes.py:3731-3751).
    Host arithmetic: ``nlevel`` numbers per climate run, nothing for a GPU."""
    x = pressure / pm
    return np.exp(1.0 + hratio * np.log(x) - x ** hratio)


def tidal_flux(T_e, nlevel, pressure, col_den, InjectionBundle):
    """Tidal (internal + injected) flux at every level (reference ``fluxes.tidal_flux``, fluxes.py:3671-3729):
    ``-sigma T_e**4`` plus the running sum of the injected energy (a Chapman profile times the column density, or the
    caller's ``beam_profile``), rescaled to the total ``wave_in`` / ``sum(beam_profile)``.  ``InjectionBundle`` carries
    ``inject_beam, beam_profile, pm, hratio, wave_in`` as in the reference.  Host arithmetic (a recurrence over
    ``nlevel`` numbers, in the reference's operation order); like the reference, levels 0 and 1 stay at zero before the
    rescaling and a profile that injects nothing divides by zero."""
    sigma_sb = 0.56687e-4
    tide = -sigma_sb * (T_e ** 4)
    T_tot = 0.0
    tidal = np.zeros(nlevel)
    if InjectionBundle.inject_beam == True:  # noqa: E712  (the reference's test: a numpy bool counts)
        beam = InjectionBundle.beam_profile
        for j in range(2, nlevel):
            tidal[j] = tidal[j - 1] - beam[j]
            T_tot += tidal[j] - tidal[j - 1]
        total = np.sum(beam)
    else:
        for j in range(2, nlevel):
            tidal[j] = tidal[j - 1] - chapman(pressure[j], InjectionBundle.pm, InjectionBundle.hratio) * col_den[j - 1]
            T_tot += tidal[j] - tidal[j - 1]
        total = InjectionBundle.wave_in
    return (tidal * total / T_tot) + tide - (tidal[-1] * total / T_tot)
