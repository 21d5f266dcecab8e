"""ctypes front-end of the CPU oracle (``oracle/picaso_oracle.c``).  TEST INFRASTRUCTURE.

Exposes the reference's Python signatures (``picaso/fluxes.py``, ``picaso/disco.py``) on top of the
plain-C restatement so parity tests read like calls into the reference.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product package ``picaso_amd`` never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_dp = ctypes.POINTER(ctypes.c_double)


def _host_stamp():
    """What -march=native resolved to depends on the CPU: model name + ISA flags of this host."""
    import hashlib
    try:
        with open("/proc/cpuinfo") as fh:
            lines = [ln for ln in fh.read().splitlines() if ln.startswith(("model name", "flags"))][:2]
    except OSError:
        lines = []
    import platform
    return hashlib.sha1(("\n".join(lines) + platform.machine()).encode()).hexdigest()[:16]


def build(force=False):
    so = os.path.join(_HERE, "libpicaso_oracle.so")
    stamp = so + ".host"
    srcs = [os.path.join(_HERE, f) for f in ("picaso_oracle.c", "sh_oracle.c", "sh_oracle_x80.c", "mix_oracle.c", "Makefile")]
    here = _host_stamp()
    try:
        with open(stamp) as fh:
            built_on = fh.read().strip()
    except OSError:
        built_on = ""
    if force or not os.path.exists(so) or not os.path.exists(os.path.join(_HERE, "libsh_oracle_x80.so")) or built_on != here or \
            os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "all"])
        with open(stamp, "w") as fh:
            fh.write(here + "\n")
    return so


def build_flags():
    """The compiler command line of the CPU baseline (``make flags``), for bench.py's cpu_baseline entry."""
    return subprocess.check_output(["make", "-C", _HERE, "-s", "flags"]).decode().strip()


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


_LIB_X80 = None


def lib_x80():
    """The Toon reflected / thermal restatement built with ``real = long double`` (x87 extended precision)."""
    global _LIB_X80
    if _LIB_X80 is None:
        build()
        _LIB_X80 = ctypes.CDLL(os.path.join(_HERE, "libpicaso_oracle_x80.so"))
    return _LIB_X80


_LIB_SH_X80 = None


def lib_sh_x80():
    """The spherical-harmonics restatement built with ``double = long double`` (sh_oracle_x80.c)."""
    global _LIB_SH_X80
    if _LIB_SH_X80 is None:
        build()
        _LIB_SH_X80 = ctypes.CDLL(os.path.join(_HERE, "libsh_oracle_x80.so"))
    return _LIB_SH_X80


class _Prec:
    """Array / scalar marshalling of one build of the restatement."""

    def __init__(self, dtype, cscalar, getlib):
        self.dtype, self.cs, self.lib = dtype, cscalar, getlib
        self.ptr = ctypes.POINTER(cscalar)

    def a(self, x, shape=None):
        a = np.ascontiguousarray(x, dtype=self.dtype)
        if shape is not None:
            a = np.ascontiguousarray(np.broadcast_to(a, shape))
        return a

    def p(self, a):
        return a.ctypes.data_as(self.ptr) if a is not None else None

    def per_wave(self, x, nwno):
        return self.a(np.zeros(nwno, dtype=self.dtype) + np.asarray(x, dtype=self.dtype))

    def zeros(self, shape):
        return np.zeros(shape, dtype=self.dtype)

    def out(self, a):
        return a if self.dtype is np.float64 else np.asarray(a, dtype=np.float64)


_F64 = _Prec(np.float64, ctypes.c_double, lib)
_X80 = _Prec(np.longdouble, ctypes.c_longdouble, lib_x80)
_X80_SH = _Prec(np.longdouble, ctypes.c_longdouble, lib_sh_x80)


def _a(x, shape=None):
    a = np.ascontiguousarray(x, dtype=np.float64)
    if shape is not None:
        a = np.ascontiguousarray(np.broadcast_to(a, shape))
    return a


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _per_wave(x, nwno):
    """Scalar-or-array (reference accepts both) -> (nwno) array."""
    return _a(np.zeros(nwno) + np.asarray(x, dtype=np.float64))


def _check(rc, what):
    if rc != 0:
        raise Exception("oracle %s failed with code %d" % (what, rc))


def _reflected(variant, nlevel, nwno, numg, numt, planes, surf_reflect, ubar0, ubar1, cos_theta,
               F0PI, single_phase, multi_phase, frac_a, frac_b, frac_c, constant_back,
               constant_forward, get_toa_intensity, get_lvl_flux, toon_coefficients, b_top, P=_F64):
    keep = [P.a(p) for p in planes]
    sr = P.per_wave(surf_reflect, nwno)
    f0 = P.per_wave(F0PI, nwno)
    u0, u1 = P.a(ubar0), P.a(ubar1)
    xint = P.zeros((numg, numt, nwno))
    lvl = [P.zeros((numg, numt, nlevel, nwno)) for _ in range(4)] if variant == 0 else [None] * 4
    want_lvl = bool(get_lvl_flux) and variant == 0
    cd = P.cs
    rc = P.lib().orc_reflected(
        ctypes.c_int(variant), ctypes.c_int(nlevel), ctypes.c_int(nwno), ctypes.c_int(numg),
        ctypes.c_int(numt), *[P.p(k) for k in keep], P.p(sr), P.p(u0), P.p(u1),
        cd(cos_theta), P.p(f0), ctypes.c_int(single_phase), ctypes.c_int(multi_phase),
        cd(frac_a), cd(frac_b), cd(frac_c), cd(constant_back), cd(constant_forward),
        ctypes.c_int(get_toa_intensity), ctypes.c_int(get_lvl_flux),
        ctypes.c_int(toon_coefficients), cd(b_top), P.p(xint),
        *[P.p(l) if want_lvl else None for l in lvl])
    _check(rc, "reflected")
    return P.out(xint), [P.out(l) if l is not None else None for l in lvl]


def get_reflected_1d(nlevel, wno, nwno, numg, numt, dtau, tau, w0, cosb, gcos2, ftau_cld, ftau_ray,
                     dtau_og, tau_og, w0_og, cosb_og, surf_reflect, ubar0, ubar1, cos_theta, F0PI,
                     single_phase, multi_phase, frac_a, frac_b, frac_c, constant_back,
                     constant_forward, get_toa_intensity=1, get_lvl_flux=0, toon_coefficients=0,
                     b_top=0, x80=False):
    """Signature of reference ``fluxes.get_reflected_1d`` (fluxes.py:1010-1015).  ``x80=True``: the same expressions
    evaluated in x87 extended precision (results rounded to float64): how far the reference's own fp64 rounding
    moves an element is ``|get_reflected_1d(...) - get_reflected_1d(..., x80=True)|``."""
    xint, lvl = _reflected(0, nlevel, nwno, numg, numt,
                           (dtau, tau, w0, cosb, gcos2, ftau_cld, ftau_ray, dtau_og, tau_og, w0_og,
                            cosb_og), surf_reflect, ubar0, ubar1, cos_theta, F0PI, single_phase,
                           multi_phase, frac_a, frac_b, frac_c, constant_back, constant_forward,
                           get_toa_intensity, get_lvl_flux, toon_coefficients, b_top, P=_X80 if x80 else _F64)
    return xint, tuple(lvl)


def get_reflected_3d(nlevel, wno, nwno, numg, numt, dtau_3d, tau_3d, w0_3d, cosb_3d, gcos2_3d,
                     ftau_cld_3d, ftau_ray_3d, dtau_og_3d, tau_og_3d, w0_og_3d, cosb_og_3d,
                     surf_reflect, ubar0, ubar1, cos_theta, F0PI, single_phase, multi_phase, frac_a,
                     frac_b, frac_c, constant_back, constant_forward):
    """Signature of reference ``fluxes.get_reflected_3d`` (fluxes.py:355-358)."""
    xint, _ = _reflected(1, nlevel, nwno, numg, numt,
                         (dtau_3d, tau_3d, w0_3d, cosb_3d, gcos2_3d, ftau_cld_3d, ftau_ray_3d,
                          dtau_og_3d, tau_og_3d, w0_og_3d, cosb_og_3d), surf_reflect, ubar0, ubar1,
                         cos_theta, F0PI, single_phase, multi_phase, frac_a, frac_b, frac_c,
                         constant_back, constant_forward, 1, 0, 0, 0.0)
    return xint


def _thermal(variant, nlevel, wno, nwno, numg, numt, tlevel, dtau, w0, cosb, plevel, ubar1,
             surf_reflect, hard_surface, dwno, calc_type, P=_F64):
    wno_, tl, dt, w0_, cb, pl, u1 = (P.a(wno), P.a(tlevel), P.a(dtau), P.a(w0), P.a(cosb), P.a(plevel),
                                     P.a(ubar1))
    sr = P.per_wave(surf_reflect, nwno)
    dw = P.per_wave(dwno, nwno)
    out = P.zeros((numg, numt, nwno))
    lvl = [P.zeros((numg, numt, nlevel, nwno)) for _ in range(4)] if variant == 0 else [None] * 4
    rc = P.lib().orc_thermal(
        ctypes.c_int(variant), ctypes.c_int(nlevel), P.p(wno_), ctypes.c_int(nwno),
        ctypes.c_int(numg), ctypes.c_int(numt), P.p(tl), P.p(dt), P.p(w0_), P.p(cb), P.p(pl), P.p(u1),
        P.p(sr), ctypes.c_int(int(hard_surface)), P.p(dw), ctypes.c_int(calc_type), P.p(out),
        *[P.p(l) for l in lvl])
    _check(rc, "thermal")
    return P.out(out), [P.out(l) if l is not None else None for l in lvl]


def get_thermal_1d(nlevel, wno, nwno, numg, numt, tlevel, dtau, w0, cosb, plevel, ubar1,
                   surf_reflect, hard_surface, dwno, calc_type, x80=False):
    """Signature of reference ``fluxes.get_thermal_1d`` (fluxes.py:1683-1684); ``x80`` as in ``get_reflected_1d``."""
    out, lvl = _thermal(0, nlevel, wno, nwno, numg, numt, tlevel, dtau, w0, cosb, plevel, ubar1,
                        surf_reflect, hard_surface, dwno, calc_type, P=_X80 if x80 else _F64)
    return out, tuple(lvl)


def get_thermal_3d(nlevel, wno, nwno, numg, numt, tlevel_3d, dtau_3d, w0_3d, cosb_3d, plevel_3d,
                   ubar1, surf_reflect, hard_surface):
    """Signature of reference ``fluxes.get_thermal_3d`` (fluxes.py:2148-2149)."""
    out, _ = _thermal(1, nlevel, wno, nwno, numg, numt, tlevel_3d, dtau_3d, w0_3d, cosb_3d,
                      plevel_3d, ubar1, surf_reflect, hard_surface, 0.0, 0)
    return out


def compress_disco(nwno, cos_theta, xint_at_top, gweight, tweight, F0PI):
    """Signature of reference ``disco.compress_disco`` (disco.py:118)."""
    x, gw, tw = _a(xint_at_top), _a(gweight), _a(tweight)
    f0 = _per_wave(F0PI, nwno)
    out = np.zeros(nwno)
    lib().orc_compress_disco(ctypes.c_int(nwno), ctypes.c_double(cos_theta), _p(x), _p(gw),
                             ctypes.c_int(len(gw)), _p(tw), ctypes.c_int(len(tw)), _p(f0), _p(out))
    return out


def compress_thermal(nwno, flux_at_top, gweight, tweight):
    """Signature of reference ``disco.compress_thermal`` (disco.py:152); 3-D or 4-D input."""
    x, gw, tw = _a(flux_at_top), _a(gweight), _a(tweight)
    inner = x.shape[2:]
    out = np.zeros(inner)
    lib().orc_compress_thermal(ctypes.c_size_t(int(np.prod(inner))), _p(x), _p(gw),
                               ctypes.c_int(len(gw)), _p(tw), ctypes.c_int(len(tw)), _p(out))
    return out


def get_reflected_SH(nlevel, nwno, numg, numt, dtau, tau, w0, cosb, ftau_cld, ftau_ray, f_deltaM,
                     dtau_og, tau_og, w0_og, cosb_og, surf_reflect, ubar0, ubar1, cos_theta, F0PI,
                     w_single_form, w_multi_form, psingle_form, w_single_rayleigh, w_multi_rayleigh,
                     psingle_rayleigh, frac_a, frac_b, frac_c, constant_back, constant_forward,
                     stream, b_top=0, flx=0, single_form=0, x80=False):
    """Signature of reference ``fluxes.get_reflected_SH`` (fluxes.py:2675-2679).  Like the
    reference, the TTHG branch multiplies ``f_deltaM`` IN PLACE once per angle (fluxes.py:2823-2824)
    -- the caller's array is modified when it is a float64 C-contiguous array.  ``x80=True``: the same source in x87
    extended precision (``sh_oracle_x80.c``; ``f_deltaM`` is then never written back): ``|fp64 - x80|`` is how far the
    reference's own fp64 rounding moves an element."""
    P = _X80_SH if x80 else _F64
    keep = [P.a(p) for p in (dtau, tau, w0, cosb, ftau_cld, ftau_ray)]
    fd = f_deltaM if (not x80 and isinstance(f_deltaM, np.ndarray) and f_deltaM.dtype == np.float64
                      and f_deltaM.flags.c_contiguous) else P.a(f_deltaM).copy()
    og = [P.a(p) for p in (dtau_og, tau_og, w0_og, cosb_og)]
    sr, f0 = P.per_wave(surf_reflect, nwno), P.per_wave(F0PI, nwno)
    u0, u1 = P.a(ubar0), P.a(ubar1)
    xint = P.zeros((numg, numt, nwno))
    flux = P.zeros((numg, numt, stream * nlevel, nwno))
    ci, cd, p = ctypes.c_int, P.cs, P.p
    rc = P.lib().orc_reflected_SH(
        ci(nlevel), ci(nwno), ci(numg), ci(numt), *[p(k) for k in keep], p(fd), *[p(k) for k in og],
        p(sr), p(u0), p(u1), cd(cos_theta), p(f0), ci(w_single_form), ci(w_multi_form),
        ci(psingle_form), ci(w_single_rayleigh), ci(w_multi_rayleigh), ci(psingle_rayleigh),
        cd(frac_a), cd(frac_b), cd(frac_c), cd(constant_back), cd(constant_forward), ci(stream),
        cd(b_top), ci(single_form), p(xint), p(flux) if flx else None)
    _check(rc, "reflected_SH")
    return P.out(xint), P.out(flux)


def get_thermal_SH(nlevel, wno, nwno, numg, numt, tlevel, dtau, tau, w0, cosb, dtau_og, tau_og,
                   w0_og, w0_no_raman, cosb_og, plevel, ubar1, surf_reflect, stream, hard_surface,
                   flx=0):
    """Signature of reference ``fluxes.get_thermal_SH`` (fluxes.py:2979-2981)."""
    if flx:
        raise Exception("oracle: flx=1 is broken in the reference (fluxes.py:3102) and not restated")
    arrs = [_a(p) for p in (wno, tlevel, dtau, tau, w0, cosb, cosb_og, plevel, ubar1)]
    sr = _per_wave(surf_reflect, nwno)
    xint = np.zeros((numg, numt, nwno))
    ci = ctypes.c_int
    wno_, tl, dt, ta, w0_, cb, cbo, pl, u1 = arrs
    rc = lib().orc_thermal_SH(ci(nlevel), _p(wno_), ci(nwno), ci(numg), ci(numt), _p(tl), _p(dt),
                              _p(ta), _p(w0_), _p(cb), _p(cbo), _p(pl), _p(u1), _p(sr), ci(stream),
                              ci(int(hard_surface)), _p(xint))
    _check(rc, "thermal_SH")
    return xint, np.zeros((numg, numt, stream * nlevel, nwno))


def get_transit_1d(z, dz, nlevel, nwno, rstar, mmw, k_b, amu, player, tlayer, colden, DTAU):
    """Signature of reference ``fluxes.get_transit_1d`` (fluxes.py:2582-2583)."""
    out = np.zeros(nwno)
    lib().orc_get_transit_1d(_p(_a(z)), _p(_a(dz)), ctypes.c_int(nlevel), ctypes.c_int(nwno),
                             ctypes.c_double(rstar), _p(_a(mmw)), ctypes.c_double(k_b),
                             ctypes.c_double(amu), _p(_a(player)), _p(_a(tlayer)), _p(_a(colden)),
                             _p(_a(DTAU)), _p(out))
    return out


def mix_all_gases_gasesfly(kappas, mixes, gauss_pts, gauss_wts, indices):
    """Signature of reference ``deq_chem.mix_all_gases_gasesfly`` (deq_chem.py:333-384): ``kappas`` a list
    of ``(npres, ntemp, nwno, nk)`` ln(kappa) tables, ``mixes`` a list of per-layer mixing ratios,
    ``indices`` = [p_low, p_hi, t_low, t_hi] per layer.  Returns ``(nlayer, nwno, nk, 4)``."""
    ks = [_a(k) for k in kappas]
    npres, ntemp, nwno, nk = ks[0].shape
    mx = _a(np.stack([np.asarray(m, dtype=float) for m in mixes]))
    idx = np.ascontiguousarray(indices, dtype=np.int32)
    nlayer = idx.shape[1]
    ptrs = (_dp * len(ks))(*[_p(k) for k in ks])
    out = np.zeros((nlayer, nwno, nk, 4))
    _check(lib().orc_mix_all_gases_gasesfly(
        ctypes.c_int(len(ks)), ptrs, ctypes.c_int(npres), ctypes.c_int(ntemp), ctypes.c_int(nwno),
        ctypes.c_int(nk), _p(mx), _p(_a(gauss_pts)), _p(_a(gauss_wts)),
        idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.c_int(nlayer), _p(out)), "mix_all_gases_gasesfly")
    return out


def blackbody(t, w):
    """Planck function per unit wavelength, cgs, (ntemp, nwave) -- restates fluxes.blackbody, reference
    picaso/fluxes.py:1660-1680 (numpy: the reference's own expression under numba)."""
    h, c, k = 6.62607004e-27, 2.99792458e+10, 1.38064852e-16
    t, w = np.atleast_1d(np.asarray(t, float)), np.atleast_1d(np.asarray(w, float))
    with np.errstate(over="ignore"):
        return ((2.0 * h * c ** 2.0) / (w ** 5.0)) * (1.0 / (np.exp((h * c) / np.outer(t, w * k)) - 1.0))


def blackbody_integrated(T, wave, dwave):
    """Three-point bin mean of the wavenumber Planck function, (ntemp, nwave) -- restates fluxes.blackbody_integrated,
    reference picaso/fluxes.py:1609-1658 (nbb = 1: wave - dwave/2, wave, wave + dwave/2, summed in that order)."""
    h, c, k = 6.62607004e-27, 2.99792458e+10, 1.38064852e-16
    c1, c2 = 2 * h * c ** 2, h * c / k
    T, wave, dwave = (np.atleast_1d(np.asarray(x, float)) for x in (T, wave, dwave))
    s = np.zeros((T.size, wave.size))
    with np.errstate(over="ignore"):
        for kk in (-1, 0, 1):
            wn = wave + kk * dwave / 2.0
            s += c1 * (wn ** 3) / (np.exp(c2 * wn[None, :] / T[:, None]) - 1)
    return s / 3.0
