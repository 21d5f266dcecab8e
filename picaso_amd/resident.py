"""HBM-resident form of the hot path: planes stay on the GPU across calls.

Used by ``bench.py``, by wavelength-sharded multi-GPU runs and by any caller that evaluates many
geometries / options on the same atmosphere.  Thin wrappers over the ``*_dev`` entry points of
``include/picaso_hip.h``; every plane argument is a ``DeviceArray`` (or a raw device address).
"""
import ctypes

import numpy as np

from ._lib import check, f64, load, ptr
from .device import DeviceArray

_ci, _cd = ctypes.c_int, ctypes.c_double
REFLECTED_PLANES = ("dtau", "tau", "w0", "cosb", "gcos2", "ftau_cld", "ftau_ray", "dtau_og",
                    "tau_og", "w0_og", "cosb_og")


def _addr(x):
    if x is None:
        return None
    return ptr(x.addr if isinstance(x, DeviceArray) else x)


def upload_scene(scene, keys, w_lo=None, w_hi=None, ctx=None):
    """Upload the named (rows, nwno) planes / (nwno) vectors of a host scene dict, optionally only
    the wavelength shard [w_lo, w_hi)."""
    out = {}
    for k in keys:
        a = f64(scene[k])
        if w_lo is None:
            out[k] = DeviceArray.from_host(a, ctx)
        elif a.ndim == 1:
            out[k] = DeviceArray.from_host(a[w_lo:w_hi], ctx)
        else:
            out[k] = DeviceArray.from_host_columns(a, w_lo, w_hi, ctx)
    return out


def reflected_can_derive(nlevel, nwno, numg, numt, ubar0, ubar1, cos_theta, single_phase, multi_phase, frac_c,
                         toon_coefficients=0, get_lvl_flux=False):
    """True when ``reflected_1d`` / ``reflected_1d_batch`` with these arguments may be handed ``None`` for the planes the
    kernels re-derive (``tau``, ``tau_og``, ``gcos2``; for an atmosphere without cloud everything but ``dtau`` and
    ``w0``): ``picaso_reflected_1d_can_derive``."""
    u0, u1 = f64(ubar0, (numg, numt)), f64(ubar1, (numg, numt))
    return bool(load().picaso_reflected_1d_can_derive(
        _ci(nlevel), ctypes.c_long(nwno), _ci(numg), _ci(numt), ptr(u0), ptr(u1), _cd(cos_theta), _ci(single_phase),
        _ci(multi_phase), _cd(frac_c), _ci(toon_coefficients), _ci(1 if get_lvl_flux else 0)))


def reflected_1d(ctx, nlevel, nwno, numg, numt, planes, surf_reflect, ubar0, ubar1, cos_theta,
                 F0PI, single_phase, multi_phase, frac_a, frac_b, frac_c, constant_back,
                 constant_forward, xint_at_top, toon_coefficients=0, b_top=0.0, gweight=None,
                 tweight=None, albedo=None, plane_pitch=None):
    """Asynchronous ``get_reflected_1d`` on resident planes (+ optional fused ``compress_disco``).
    ``planes`` maps the 11 reference plane names to DeviceArrays; outputs are DeviceArrays (or raw
    device addresses)."""
    u0, u1 = f64(ubar0, (numg, numt)), f64(ubar1, (numg, numt))
    gw = f64(gweight) if gweight is not None else None
    tw = f64(tweight) if tweight is not None else None
    pitch = nwno if plane_pitch is None else plane_pitch
    check(load().picaso_get_reflected_1d_dev(
        ctx, _ci(nlevel), _ci(nwno), ctypes.c_long(pitch), _ci(numg), _ci(numt),
        *[_addr(planes.get(k)) for k in REFLECTED_PLANES], _addr(surf_reflect), ptr(u0), ptr(u1),
        _cd(cos_theta), _addr(F0PI), _ci(single_phase), _ci(multi_phase), _cd(frac_a), _cd(frac_b),
        _cd(frac_c), _cd(constant_back), _cd(constant_forward), _ci(1), _ci(0),
        _ci(toon_coefficients), _cd(b_top), _addr(xint_at_top), None, None, None, None,
        ptr(gw) if gw is not None else None, ptr(tw) if tw is not None else None, _addr(albedo)),
        ctx)


def thermal_1d(ctx, nlevel, wno, nwno, numg, numt, tlevel, dtau, w0, cosb, plevel, ubar1,
               surf_reflect, hard_surface, flux_at_top, dwno=None, calc_type=0, gweight=None,
               tweight=None, flux_disk=None, plane_pitch=None, lvl_fluxes=None):
    """Asynchronous ``get_thermal_1d`` on resident planes (+ optional fused ``compress_thermal``).
    ``wno``/``dwno``/``surf_reflect`` and the planes are device-resident; tlevel/plevel host."""
    u1 = f64(ubar1, (numg, numt))
    tl, pl = f64(tlevel), f64(plevel)
    gw = f64(gweight) if gweight is not None else None
    tw = f64(tweight) if tweight is not None else None
    pitch = nwno if plane_pitch is None else plane_pitch
    check(load().picaso_get_thermal_1d_dev(
        ctx, _ci(nlevel), _addr(wno), _ci(nwno), ctypes.c_long(pitch), _ci(numg), _ci(numt),
        ptr(tl), _addr(dtau), _addr(w0), _addr(cosb), ptr(pl), ptr(u1), _addr(surf_reflect),
        _ci(int(hard_surface)), _addr(dwno), _ci(calc_type), _addr(flux_at_top),
        *[_addr(x) for x in (lvl_fluxes if lvl_fluxes is not None else [None] * 4)],
        ptr(gw) if gw is not None else None, ptr(tw) if tw is not None else None,
        _addr(flux_disk)), ctx)


def _ptr_array(items):
    """Host array of device addresses (``const double *const *`` of the batched entry points)."""
    arr = (ctypes.c_void_p * len(items))(*[(x.addr if isinstance(x, DeviceArray) else x) for x in items])   # None: NULL
    return arr, ctypes.cast(arr, ctypes.POINTER(ctypes.POINTER(ctypes.c_double)))


def _per_spectrum(x, nspec):
    """One entry for all spectra or a list of ``nspec``."""
    if isinstance(x, (list, tuple)):
        if len(x) != nspec:
            raise Exception("batched call: expected %d per-spectrum entries, got %d" % (nspec, len(x)))
        return list(x)
    return [x] * nspec


def reflected_1d_batch(ctx, nlevel, nwno, numg, numt, planes, surf_reflect, ubar0, ubar1, cos_theta, F0PI,
                       single_phase, multi_phase, frac_a, frac_b, frac_c, constant_back, constant_forward,
                       xint_at_top, toon_coefficients=0, b_top=0.0, gweight=None, tweight=None, albedo=None,
                       plane_pitch=None):
    """``len(planes)`` spectra of one shape and option set in ONE launch (``picaso_get_reflected_1d_batch_dev``):
    ``planes`` is a list of plane dictionaries as ``reflected_1d`` takes them (entries may be the same
    dictionary: one atmosphere under several geometries), ``xint_at_top`` / ``albedo`` lists of DeviceArrays,
    ``surf_reflect`` / ``F0PI`` one DeviceArray for all or a list.  Geometry: ``ubar0`` / ``ubar1`` ``(numg, numt)``
    and a scalar ``cos_theta`` for all spectra, or ``(nspec, numg, numt)`` and ``nspec`` values.  Spectrum ``s``
    of the result is bit-identical to ``reflected_1d`` on its own arguments."""
    nspec = len(planes)
    u0, u1 = np.asarray(ubar0, dtype=np.float64), np.asarray(ubar1, dtype=np.float64)
    ngeom = nspec if u0.ndim == 3 else 1
    shape = (nspec, numg, numt) if ngeom > 1 else (numg, numt)
    u0, u1 = f64(u0, shape), f64(u1, shape)
    ct = f64(np.zeros(ngeom) + np.asarray(cos_theta, dtype=np.float64), (ngeom,))
    gw = f64(gweight) if gweight is not None else None
    tw = f64(tweight) if tweight is not None else None
    pitch = nwno if plane_pitch is None else plane_pitch
    keep, cols = [], []
    for k in REFLECTED_PLANES:
        a, p = _ptr_array([pl.get(k) for pl in planes])
        keep.append(a)
        cols.append(p)
    a_rs, p_rs = _ptr_array(_per_spectrum(surf_reflect, nspec))
    a_f0, p_f0 = _ptr_array(_per_spectrum(F0PI, nspec))
    a_x, p_x = _ptr_array(_per_spectrum(xint_at_top, nspec))
    fuse = albedo is not None and gw is not None and tw is not None
    a_al, p_al = _ptr_array(_per_spectrum(albedo, nspec)) if fuse else (None, None)
    check(load().picaso_get_reflected_1d_batch_dev(
        ctx, _ci(nspec), _ci(nlevel), _ci(nwno), ctypes.c_long(pitch), _ci(numg), _ci(numt), *cols, p_rs,
        _ci(ngeom), ptr(u0), ptr(u1), ptr(ct), p_f0, _ci(single_phase), _ci(multi_phase), _cd(frac_a), _cd(frac_b),
        _cd(frac_c), _cd(constant_back), _cd(constant_forward), _ci(toon_coefficients), _cd(b_top), p_x,
        ptr(gw) if fuse else None, ptr(tw) if fuse else None, p_al), ctx)


def thermal_1d_batch(ctx, nlevel, wno, nwno, numg, numt, tlevel, dtau, w0, cosb, plevel, ubar1, surf_reflect,
                     hard_surface, flux_at_top, dwno=None, calc_type=0, gweight=None, tweight=None, flux_disk=None,
                     plane_pitch=None):
    """``len(dtau)`` thermal spectra in ONE launch (``picaso_get_thermal_1d_batch_dev``): ``tlevel`` / ``plevel``
    host ``(nspec, nlevel)`` (or ``(nlevel,)`` for all), ``dtau`` / ``w0`` / ``cosb`` / ``flux_at_top`` /
    ``flux_disk`` lists of DeviceArrays, ``surf_reflect`` one DeviceArray or a list, ``wno`` / ``dwno`` shared;
    ``ubar1`` ``(numg, numt)`` or ``(nspec, numg, numt)``.  Bit-identical per spectrum to ``thermal_1d``."""
    nspec = len(dtau)
    u1 = np.asarray(ubar1, dtype=np.float64)
    ngeom = nspec if u1.ndim == 3 else 1
    u1 = f64(u1, (nspec, numg, numt) if ngeom > 1 else (numg, numt))
    tl, pl = f64(tlevel, (nspec, nlevel)), f64(plevel, (nspec, nlevel))
    gw = f64(gweight) if gweight is not None else None
    tw = f64(tweight) if tweight is not None else None
    pitch = nwno if plane_pitch is None else plane_pitch
    a_dt, p_dt = _ptr_array(list(dtau))
    a_w0, p_w0 = _ptr_array(_per_spectrum(w0, nspec))
    a_cb, p_cb = _ptr_array(_per_spectrum(cosb, nspec))
    a_rs, p_rs = _ptr_array(_per_spectrum(surf_reflect, nspec))
    a_fx, p_fx = _ptr_array(_per_spectrum(flux_at_top, nspec))
    fuse = flux_disk is not None and gw is not None and tw is not None
    a_fd, p_fd = _ptr_array(_per_spectrum(flux_disk, nspec)) if fuse else (None, None)
    check(load().picaso_get_thermal_1d_batch_dev(
        ctx, _ci(nspec), _ci(nlevel), _addr(wno), _ci(nwno), ctypes.c_long(pitch), _ci(numg), _ci(numt), ptr(tl),
        p_dt, p_w0, p_cb, ptr(pl), _ci(ngeom), ptr(u1), p_rs, _ci(int(hard_surface)), _addr(dwno), _ci(calc_type),
        p_fx, ptr(gw) if fuse else None, ptr(tw) if fuse else None, p_fd), ctx)


def reflected_1d_ck(ctx, nlevel, nwno, ngauss, numg, numt, planes, surf_reflect, ubar0, ubar1,
                    cos_theta, F0PI, single_phase, multi_phase, frac_a, frac_b, frac_c,
                    constant_back, constant_forward, gauss_wts, xint_at_top, toon_coefficients=0,
                    b_top=0.0, gweight=None, tweight=None, albedo=None, get_toa_intensity=1,
                    lvl_fluxes=None):
    """The reference's ``for ig in range(ngauss)`` loop around ``get_reflected_1d``
    (justdoit.py:256-313) as one launch: planes are ``(nlayer|nlevel, nwno, ngauss)`` DeviceArrays
    exactly as ``compute_opacity`` lays them out, ``xint_at_top`` is the Gauss-weighted
    ``(numg,numt,nwno)`` result; ``lvl_fluxes`` = four ``(numg,numt,nlevel,nwno)`` DeviceArrays
    (flux_minus, flux_plus, flux_minus_mdpt, flux_plus_mdpt) turns ``get_lvl_flux`` on."""
    u0, u1 = f64(ubar0, (numg, numt)), f64(ubar1, (numg, numt))
    gw = f64(gweight) if gweight is not None else None
    tw = f64(tweight) if tweight is not None else None
    wts = f64(gauss_wts, (ngauss,))
    lv = list(lvl_fluxes) if lvl_fluxes is not None else [None] * 4
    check(load().picaso_get_reflected_1d_ck_dev(
        ctx, _ci(nlevel), _ci(nwno), _ci(ngauss), _ci(numg), _ci(numt),
        *[_addr(planes[k]) for k in REFLECTED_PLANES], _addr(surf_reflect), ptr(u0), ptr(u1),
        _cd(cos_theta), _addr(F0PI), _ci(single_phase), _ci(multi_phase), _cd(frac_a), _cd(frac_b),
        _cd(frac_c), _cd(constant_back), _cd(constant_forward), _ci(int(get_toa_intensity)),
        _ci(1 if lvl_fluxes is not None else 0), _ci(toon_coefficients), _cd(b_top), ptr(wts),
        _addr(xint_at_top), *[_addr(x) for x in lv], ptr(gw) if gw is not None else None,
        ptr(tw) if tw is not None else None, _addr(albedo)), ctx)


def thermal_1d_ck(ctx, nlevel, wno, nwno, ngauss, numg, numt, tlevel, dtau, w0, cosb, plevel, ubar1,
                  surf_reflect, hard_surface, gauss_wts, flux_at_top, dwno=None, calc_type=0,
                  gweight=None, tweight=None, flux_disk=None, lvl_fluxes=None):
    """The ``ngauss`` loop around ``get_thermal_1d`` (justdoit.py:328-380) as one launch; planes
    ``(nlayer, nwno, ngauss)``, result Gauss-weighted ``(numg,numt,nwno)``; optional Gauss-weighted
    level fluxes as in ``reflected_1d_ck``."""
    u1 = f64(ubar1, (numg, numt))
    tl, pl = f64(tlevel), f64(plevel)
    gw = f64(gweight) if gweight is not None else None
    tw = f64(tweight) if tweight is not None else None
    wts = f64(gauss_wts, (ngauss,))
    lv = list(lvl_fluxes) if lvl_fluxes is not None else [None] * 4
    check(load().picaso_get_thermal_1d_ck_dev(
        ctx, _ci(nlevel), _addr(wno), _ci(nwno), _ci(ngauss), _ci(numg), _ci(numt), ptr(tl),
        _addr(dtau), _addr(w0), _addr(cosb), ptr(pl), ptr(u1), _addr(surf_reflect),
        _ci(int(hard_surface)), _addr(dwno), _ci(calc_type), ptr(wts), _addr(flux_at_top),
        *[_addr(x) for x in lv], ptr(gw) if gw is not None else None,
        ptr(tw) if tw is not None else None, _addr(flux_disk)), ctx)


def thermal_1d_ck_tbatch(ctx, nlevel, wno, nwno, ngauss, numg, numt, tlevels, dtau, w0, cosb, plevel, ubar1, surf_reflect,
                         hard_surface, gauss_wts, gweight, tweight, disk4, dwno=None, calc_type=0):
    """The level fluxes of ``thermal_1d_ck`` + ``compress_thermal`` for every row of ``tlevels`` ``(nitem, nlevel)``
    over ONE set of planes, in one launch sequence (``picaso_get_thermal_1d_ck_tbatch_dev``): ``disk4`` is a
    DeviceArray ``(4, nlevel, nitem*nwno)`` -- flux_minus, flux_plus, flux_minus_mdpt, flux_plus_mdpt, Gauss-weighted and
    disk-integrated, column = profile * nwno + wavelength."""
    tl = f64(tlevels)
    nitem = tl.shape[0]
    tl = f64(tl, (nitem, nlevel))
    check(load().picaso_get_thermal_1d_ck_tbatch_dev(
        ctx, _ci(nitem), _ci(nlevel), _addr(wno), _ci(nwno), _ci(ngauss), _ci(numg), _ci(numt), ptr(tl), _addr(dtau),
        _addr(w0), _addr(cosb), ptr(f64(plevel, (nlevel,))), ptr(f64(ubar1, (numg, numt))), _addr(surf_reflect),
        _ci(int(hard_surface)), _addr(dwno), _ci(calc_type), ptr(f64(gauss_wts, (ngauss,))), ptr(f64(gweight)),
        ptr(f64(tweight)), _addr(disk4)), ctx)


def transit_1d_ck(ctx, z, dz, nlevel, nwno, ngauss, rstar, mmw, k_b, amu, player, tlayer, colden, dtau,
                  gauss_wts, rprs2):
    """The transmission branch's correlated-k loop (reference justdoit.py:388-405) on a resident
    ``DTAU_OG`` plane ``(nlevel-1, nwno*ngauss)``, Gauss index fastest; ``rprs2`` (nwno) DeviceArray."""
    wts = f64(gauss_wts, (ngauss,))
    check(load().picaso_get_transit_1d_ck_dev(
        ctx, ptr(f64(z, (nlevel,))), ptr(f64(dz, (nlevel,))), _ci(nlevel), _ci(nwno), _ci(ngauss),
        _cd(rstar), ptr(f64(mmw, (nlevel - 1,))), _cd(k_b), _cd(amu), ptr(f64(player)), ptr(f64(tlayer)),
        ptr(f64(colden, (nlevel - 1,))), _addr(dtau), ptr(wts), _addr(rprs2)), ctx)


def mix_all_gases_gasesfly(ctx, kappas, mixes, gauss_pts, gauss_wts, indices, out=None):
    """On-the-fly correlated-k gas mixing, reference ``deq_chem.mix_all_gases_gasesfly``
    (deq_chem.py:333-384).  ``kappas``: list of DeviceArrays ``(npres, ntemp, nwno, ngauss)`` of
    ln(kappa), one per gas; ``mixes``: list of per-layer mixing ratios; ``indices`` =
    [p_low, p_hi, t_low, t_hi].  Returns a DeviceArray ``(nlayer, 4, nwno, ngauss)``: the reference's
    ``(nlayer, nwno, ngauss, 4)`` array with the neighbour axis moved forward."""
    npres, ntemp, nwno, nk = kappas[0].shape
    for k in kappas:
        if tuple(k.shape) != (npres, ntemp, nwno, nk):
            raise Exception("mix_all_gases_gasesfly: all gas tables must share one (npres, ntemp, nwno, ngauss)")
    mx = f64(np.stack([np.asarray(m, dtype=float) for m in mixes]))
    idx = np.ascontiguousarray(indices, dtype=np.int32)
    nlayer = idx.shape[1]
    if idx.shape[0] != 4 or mx.shape != (len(kappas), nlayer):
        raise Exception("mix_all_gases_gasesfly: indices must be (4, nlayer) and mixes (ngas, nlayer)")
    if out is None:
        out = DeviceArray((nlayer, 4, nwno, nk), ctx)
    ptrs = (ctypes.c_void_p * len(kappas))(*[k.addr for k in kappas])
    check(load().picaso_mix_all_gases_gasesfly_dev(
        ctx, _ci(len(kappas)), ctypes.cast(ptrs, ctypes.POINTER(ctypes.POINTER(ctypes.c_double))), _ci(npres),
        _ci(ntemp), _ci(nwno), _ci(nk), ptr(mx), ptr(f64(gauss_pts, (nk,))), ptr(f64(gauss_wts, (nk,))),
        idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _ci(nlayer), _addr(out)), ctx)
    return out


def axpby(ctx, a, x, b, y, out):
    """``out = a*x + b*y`` on DeviceArrays of equal size (patchy-cloud blend, justdoit.py:300-305)."""
    n = int(np.prod(x.shape))
    check(load().picaso_axpby_dev(ctx, ctypes.c_size_t(n), _cd(a), _addr(x), _cd(b), _addr(y),
                                  _addr(out)), ctx)


def trapz(ctx, n, d, y, out, mult=None, reverse=False):
    """``np.trapezoid(y * mult, x)`` with ``d = diff(x)`` resident, summed in numpy's own pairwise order (same bits);
    ``reverse``: over ``y[::-1]`` (and ``mult[::-1]``).  ``out``: a DeviceArray / address of one double
    (``picaso_trapz_dev``; the Bond-albedo and effective-temperature integrals of justdoit.py:552-599)."""
    check(load().picaso_trapz_dev(ctx, ctypes.c_long(int(n)), _addr(d), _addr(y), _addr(mult), _ci(1 if reverse else 0),
                                  _addr(out)), ctx)


def compress_disco(ctx, nwno, cos_theta, xint_at_top, gweight, tweight, F0PI, albedo):
    """``disco.compress_disco`` on DeviceArrays (reference disco.py:117-149)."""
    gw, tw = f64(gweight), f64(tweight)
    check(load().picaso_compress_disco_dev(ctx, _ci(nwno), _cd(cos_theta), _addr(xint_at_top), ptr(gw),
                                           _ci(gw.size), ptr(tw), _ci(tw.size), _addr(F0PI),
                                           _addr(albedo)), ctx)


def compress_thermal(ctx, ninner, flux_at_top, gweight, tweight, flux):
    """``disco.compress_thermal`` on DeviceArrays (reference disco.py:151-181)."""
    gw, tw = f64(gweight), f64(tweight)
    check(load().picaso_compress_thermal_dev(ctx, ctypes.c_size_t(ninner), _addr(flux_at_top), ptr(gw),
                                             _ci(gw.size), ptr(tw), _ci(tw.size), _addr(flux)), ctx)


SH_PLANES = ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray", "f_deltaM", "dtau_og", "tau_og", "w0_og",
             "cosb_og")


def reflected_SH_can_derive(stream, w_single_form=0, w_multi_form=0, psingle_form=0, w_single_rayleigh=1,
                            w_multi_rayleigh=1, psingle_rayleigh=1, frac_c=2.0, single_form=0, flx=0):
    """True when ``reflected_SH`` with these options takes a cloud-free atmosphere as ``dtau`` and ``w0`` only
    (``picaso_reflected_SH_can_derive``)."""
    lib = load()
    lib.picaso_reflected_SH_can_derive.argtypes = [ctypes.c_int] * 7 + [ctypes.c_double] + [ctypes.c_int] * 2
    return bool(lib.picaso_reflected_SH_can_derive(int(stream), int(w_single_form), int(w_multi_form), int(psingle_form),
                                                   int(w_single_rayleigh), int(w_multi_rayleigh), int(psingle_rayleigh),
                                                   float(frac_c), int(single_form), int(flx)))


def reflected_SH_can_derive_levels(nlevel, plane_pitch, stream, w_single_form=0, w_multi_form=0, psingle_form=0,
                                   w_single_rayleigh=1, w_multi_rayleigh=1, psingle_rayleigh=1, frac_c=2.0, single_form=0,
                                   flx=0):
    """True when ``reflected_SH`` / ``reflected_SH_ck`` with these options may be handed ``None`` for the level planes
    ``tau`` and ``tau_og`` (both): the launch then carries the beam exponentials as running products of the layers'
    (``picaso_reflected_SH_can_derive_levels``)."""
    lib = load()
    lib.picaso_reflected_SH_can_derive_levels.argtypes = ([ctypes.c_int, ctypes.c_long] + [ctypes.c_int] * 7
                                                          + [ctypes.c_double] + [ctypes.c_int] * 2)
    return bool(lib.picaso_reflected_SH_can_derive_levels(
        int(nlevel), int(plane_pitch), int(stream), int(w_single_form), int(w_multi_form), int(psingle_form),
        int(w_single_rayleigh), int(w_multi_rayleigh), int(psingle_rayleigh), float(frac_c), int(single_form), int(flx)))


def reflected_SH(ctx, nlevel, nwno, numg, numt, planes, surf_reflect, ubar0, ubar1, cos_theta, F0PI,
                 w_single_form, w_multi_form, psingle_form, w_single_rayleigh, w_multi_rayleigh,
                 psingle_rayleigh, frac_a, frac_b, frac_c, constant_back, constant_forward, stream,
                 xint_at_top, b_top=0.0, single_form=0, compound_f_deltaM=True, gweight=None, tweight=None,
                 albedo=None, plane_pitch=None, cloud_free_above=0):
    """Asynchronous ``get_reflected_SH`` (``flx=0``) on resident planes (+ optional fused
    ``compress_disco``); ``planes`` maps ``SH_PLANES`` to DeviceArrays.  A cloud-free atmosphere may give ``dtau``
    and ``w0`` only (``reflected_SH_can_derive``): the other planes are constants, copies and running sums.
    ``cloud_free_above``: the caller's statement that the first that many layers carry no cloud in any column
    (``picaso_get_reflected_SH_top_dev``)."""
    u0, u1 = f64(ubar0, (numg, numt)), f64(ubar1, (numg, numt))
    gw = f64(gweight) if gweight is not None else None
    tw = f64(tweight) if tweight is not None else None
    pitch = nwno if plane_pitch is None else plane_pitch
    check(load().picaso_get_reflected_SH_top_dev(
        ctx, _ci(nlevel), _ci(nwno), ctypes.c_long(pitch), _ci(numg), _ci(numt),
        *[_addr(planes.get(k)) for k in SH_PLANES], _addr(surf_reflect), ptr(u0), ptr(u1), _cd(cos_theta),
        _addr(F0PI), _ci(int(w_single_form)), _ci(int(w_multi_form)), _ci(int(psingle_form)),
        _ci(int(w_single_rayleigh)), _ci(int(w_multi_rayleigh)), _ci(int(psingle_rayleigh)), _cd(frac_a),
        _cd(frac_b), _cd(frac_c), _cd(constant_back), _cd(constant_forward), _ci(int(stream)), _cd(b_top),
        _ci(0), _ci(int(single_form)), _ci(1 if compound_f_deltaM else 0), _ci(int(cloud_free_above)),
        _addr(xint_at_top), None,
        ptr(gw) if gw is not None else None, ptr(tw) if tw is not None else None, _addr(albedo)), ctx)


def reflected_SH_batch(ctx, nlevel, nwno, numg, numt, planes, surf_reflect, ubar0, ubar1, cos_theta, F0PI,
                       w_single_form, w_multi_form, psingle_form, w_single_rayleigh, w_multi_rayleigh,
                       psingle_rayleigh, frac_a, frac_b, frac_c, constant_back, constant_forward, stream,
                       xint_at_top, b_top=0.0, single_form=0, compound_f_deltaM=True, gweight=None, tweight=None,
                       albedo=None, plane_pitch=None):
    """``len(planes)`` SH spectra (``flx=0``) in ONE launch (``picaso_get_reflected_SH_batch_dev``); arguments as
    ``reflected_1d_batch`` with ``planes`` a list of ``SH_PLANES`` dictionaries.  Bit-identical per spectrum to
    ``reflected_SH``."""
    nspec = len(planes)
    u0, u1 = np.asarray(ubar0, dtype=np.float64), np.asarray(ubar1, dtype=np.float64)
    ngeom = nspec if u0.ndim == 3 else 1
    shape = (nspec, numg, numt) if ngeom > 1 else (numg, numt)
    u0, u1 = f64(u0, shape), f64(u1, shape)
    ct = f64(np.zeros(ngeom) + np.asarray(cos_theta, dtype=np.float64), (ngeom,))
    gw = f64(gweight) if gweight is not None else None
    tw = f64(tweight) if tweight is not None else None
    pitch = nwno if plane_pitch is None else plane_pitch
    keep, cols = [], []
    for k in SH_PLANES:
        a, p = _ptr_array([pl[k] for pl in planes])
        keep.append(a)
        cols.append(p)
    a_rs, p_rs = _ptr_array(_per_spectrum(surf_reflect, nspec))
    a_f0, p_f0 = _ptr_array(_per_spectrum(F0PI, nspec))
    a_x, p_x = _ptr_array(_per_spectrum(xint_at_top, nspec))
    fuse = albedo is not None and gw is not None and tw is not None
    a_al, p_al = _ptr_array(_per_spectrum(albedo, nspec)) if fuse else (None, None)
    check(load().picaso_get_reflected_SH_batch_dev(
        ctx, _ci(nspec), _ci(nlevel), _ci(nwno), ctypes.c_long(pitch), _ci(numg), _ci(numt), *cols, p_rs, _ci(ngeom),
        ptr(u0), ptr(u1), ptr(ct), p_f0, _ci(int(w_single_form)), _ci(int(w_multi_form)), _ci(int(psingle_form)),
        _ci(int(w_single_rayleigh)), _ci(int(w_multi_rayleigh)), _ci(int(psingle_rayleigh)), _cd(frac_a), _cd(frac_b),
        _cd(frac_c), _cd(constant_back), _cd(constant_forward), _ci(int(stream)), _cd(b_top), _ci(int(single_form)),
        _ci(1 if compound_f_deltaM else 0), p_x, ptr(gw) if fuse else None, ptr(tw) if fuse else None, p_al), ctx)


def reflected_3d(ctx, nlevel, nwno, numg, numt, planes, surf_reflect, ubar0, ubar1, cos_theta, F0PI,
                 single_phase, multi_phase, frac_a, frac_b, frac_c, constant_back, constant_forward,
                 xint_at_top, gweight=None, tweight=None, albedo=None):
    """``get_reflected_3d`` on resident ``(nlayer|nlevel, nwno, numg, numt)`` planes (+ optional fused
    ``compress_disco``).  Planes missing from ``planes`` are passed as NULL: the kernel re-derives them
    (picaso_hip.h: tau / tau_og / gcos2; no-cloud constants; no delta-scaling)."""
    u0, u1 = f64(ubar0, (numg, numt)), f64(ubar1, (numg, numt))
    gw = f64(gweight) if gweight is not None else None
    tw = f64(tweight) if tweight is not None else None
    check(load().picaso_get_reflected_3d_dev(
        ctx, _ci(nlevel), _ci(nwno), _ci(numg), _ci(numt), *[_addr(planes.get(k)) for k in REFLECTED_PLANES],
        _addr(surf_reflect), ptr(u0), ptr(u1), _cd(cos_theta), _addr(F0PI), _ci(single_phase),
        _ci(multi_phase), _cd(frac_a), _cd(frac_b), _cd(frac_c), _cd(constant_back),
        _cd(constant_forward), _addr(xint_at_top), ptr(gw) if gw is not None else None,
        ptr(tw) if tw is not None else None, _addr(albedo)), ctx)


def thermal_3d(ctx, nlevel, wno, nwno, numg, numt, tlevel_3d, dtau_3d, w0_3d, cosb_3d, plevel_3d, ubar1,
               surf_reflect, hard_surface, int_at_top, gweight=None, tweight=None, flux_disk=None):
    """``get_thermal_3d`` on resident ``(nlayer, nwno, numg, numt)`` planes; ``tlevel_3d`` /
    ``plevel_3d`` host ``(nlevel, numg, numt)``."""
    u1 = f64(ubar1, (numg, numt))
    tl, pl = f64(tlevel_3d, (nlevel, numg, numt)), f64(plevel_3d, (nlevel, numg, numt))
    gw = f64(gweight) if gweight is not None else None
    tw = f64(tweight) if tweight is not None else None
    check(load().picaso_get_thermal_3d_dev(
        ctx, _ci(nlevel), _addr(wno), _ci(nwno), _ci(numg), _ci(numt), ptr(tl), _addr(dtau_3d),
        _addr(w0_3d), _addr(cosb_3d), ptr(pl), ptr(u1), _addr(surf_reflect), _ci(int(hard_surface)),
        _addr(int_at_top), ptr(gw) if gw is not None else None, ptr(tw) if tw is not None else None,
        _addr(flux_disk)), ctx)


def reflected_3d_batch(ctx, nlevel, nwno, numg, numt, planes, surf_reflect, ubar0, ubar1, cos_theta, F0PI,
                       single_phase, multi_phase, frac_a, frac_b, frac_c, constant_back, constant_forward,
                       xint_at_top, gweight=None, tweight=None, albedo=None):
    """``len(planes)`` 3-D spectra (the phases of a phase curve) in ONE launch
    (``picaso_get_reflected_3d_batch_dev``): ``planes`` a list of plane dictionaries as ``reflected_3d`` takes them,
    all with the same keys (a plane family the kernel re-derives is left out of every one of them); ``ubar0`` /
    ``ubar1`` ``(nspec, numg, numt)``, ``cos_theta`` ``(nspec,)``.  Bit-identical per spectrum to ``reflected_3d``."""
    nspec = len(planes)
    keys = set(planes[0].keys())
    if any(set(pl.keys()) != keys for pl in planes):
        raise Exception("reflected_3d_batch: every spectrum must hand over the same set of planes")
    u0, u1 = f64(ubar0, (nspec, numg, numt)), f64(ubar1, (nspec, numg, numt))
    ct = f64(np.zeros(nspec) + np.asarray(cos_theta, dtype=np.float64), (nspec,))
    gw = f64(gweight) if gweight is not None else None
    tw = f64(tweight) if tweight is not None else None
    keep, cols = [], []
    for k in REFLECTED_PLANES:
        if planes[0].get(k) is None:
            cols.append(None)
            continue
        a, p = _ptr_array([pl[k] for pl in planes])
        keep.append(a)
        cols.append(p)
    a_rs, p_rs = _ptr_array(_per_spectrum(surf_reflect, nspec))
    a_f0, p_f0 = _ptr_array(_per_spectrum(F0PI, nspec))
    a_x, p_x = _ptr_array(_per_spectrum(xint_at_top, nspec))
    fuse = albedo is not None and gw is not None and tw is not None
    a_al, p_al = _ptr_array(_per_spectrum(albedo, nspec)) if fuse else (None, None)
    check(load().picaso_get_reflected_3d_batch_dev(
        ctx, _ci(nspec), _ci(nlevel), _ci(nwno), _ci(numg), _ci(numt), *cols, p_rs, ptr(u0), ptr(u1), ptr(ct), p_f0,
        _ci(single_phase), _ci(multi_phase), _cd(frac_a), _cd(frac_b), _cd(frac_c), _cd(constant_back),
        _cd(constant_forward), p_x, ptr(gw) if fuse else None, ptr(tw) if fuse else None, p_al), ctx)


def reflected_3d_fm_batch(ctx, nlevel, nwno, numg, numt, planes, surf_reflect, ubar0, ubar1, cos_theta, F0PI,
                          single_phase, multi_phase, frac_a, frac_b, frac_c, constant_back, constant_forward,
                          xint_at_top, gweight, tweight, albedo):
    """``reflected_3d_batch`` on FACET-MAJOR planes -- ``(nfacets, rows, nwno)``, what the fused gas + mixing launch
    writes for the tall atmosphere of all facets (``optics.compute_opacity_facet_major``): every facet of every spectrum
    goes into the batched launch as a spectrum of its own with one facet (its planes = one slab of the stack, its geometry
    = that facet's ``ubar0 / ubar1``; a wave then holds 64 wavelengths of one facet instead of the 64 facets of one
    wavelength), followed by the disk sum of each spectrum.  Same kernel body, same bits as the facet-fastest layout."""
    nspec, nfac = len(planes), numg * numt
    u0 = f64(ubar0, (nspec, numg, numt)).reshape(nspec * nfac, 1, 1)
    u1 = f64(ubar1, (nspec, numg, numt)).reshape(nspec * nfac, 1, 1)
    ct = np.repeat(f64(np.zeros(nspec) + np.asarray(cos_theta, dtype=np.float64), (nspec,)), nfac)
    rs, f0 = _per_spectrum(surf_reflect, nspec), _per_spectrum(F0PI, nspec)
    keys = [k for k in REFLECTED_PLANES if planes[0].get(k) is not None]
    pseudo = []
    for pl in planes:
        stride = {k: (pl[k].nbytes // nfac) for k in keys}
        pseudo += [{k: pl[k].addr + f * stride[k] for k in keys} for f in range(nfac)]
    xs = [x.addr + 8 * nwno * f for x in xint_at_top for f in range(nfac)]
    reflected_3d_batch(ctx, nlevel, nwno, 1, 1, pseudo, [r for r in rs for _ in range(nfac)], u0, u1, ct,
                       [x for x in f0 for _ in range(nfac)], single_phase, multi_phase, frac_a, frac_b, frac_c,
                       constant_back, constant_forward, xs)
    if albedo is not None and gweight is not None and tweight is not None:
        cts = np.zeros(nspec) + np.asarray(cos_theta, dtype=np.float64)
        for s in range(nspec):
            compress_disco(ctx, nwno, float(cts[s]), xint_at_top[s], gweight, tweight, f0[s], albedo[s])


def thermal_3d_fm_batch(ctx, nlevel, wno, nwno, numg, numt, tlevel_3d, dtau, w0, cosb, plevel_3d, ubar1, surf_reflect,
                        hard_surface, int_at_top, gweight, tweight, flux_disk):
    """``thermal_3d_batch`` on facet-major planes (see ``reflected_3d_fm_batch``)."""
    nspec, nfac = len(dtau), numg * numt
    u1 = f64(ubar1, (nspec, numg, numt)).reshape(nspec * nfac, 1, 1)
    tl = np.ascontiguousarray(np.moveaxis(f64(tlevel_3d, (nspec, nlevel, numg, numt)).reshape(nspec, nlevel, nfac), 2, 1))
    pl = np.ascontiguousarray(np.moveaxis(f64(plevel_3d, (nspec, nlevel, numg, numt)).reshape(nspec, nlevel, nfac), 2, 1))
    rs = _per_spectrum(surf_reflect, nspec)

    def slabs(arrs):
        return [a.addr + f * (a.nbytes // nfac) for a in arrs for f in range(nfac)]
    thermal_3d_batch(ctx, nlevel, wno, nwno, 1, 1, tl.reshape(nspec * nfac, nlevel, 1, 1), slabs(dtau), slabs(w0),
                     slabs(cosb) if cosb is not None else None, pl.reshape(nspec * nfac, nlevel, 1, 1), u1,
                     [r for r in rs for _ in range(nfac)], hard_surface,
                     [x.addr + 8 * nwno * f for x in int_at_top for f in range(nfac)])
    if flux_disk is not None and gweight is not None and tweight is not None:
        for s in range(nspec):
            compress_thermal(ctx, nwno, int_at_top[s], gweight, tweight, flux_disk[s])


def thermal_3d_batch(ctx, nlevel, wno, nwno, numg, numt, tlevel_3d, dtau_3d, w0_3d, cosb_3d, plevel_3d, ubar1,
                     surf_reflect, hard_surface, int_at_top, gweight=None, tweight=None, flux_disk=None):
    """``len(dtau_3d)`` 3-D thermal spectra in ONE launch (``picaso_get_thermal_3d_batch_dev``): ``tlevel_3d`` /
    ``plevel_3d`` host ``(nspec, nlevel, numg, numt)``, ``ubar1`` ``(nspec, numg, numt)`` or ``(numg, numt)``; the
    planes lists of DeviceArrays (``cosb_3d=None``: no cloud in any spectrum).  Bit-identical per spectrum to
    ``thermal_3d``."""
    nspec = len(dtau_3d)
    u1 = f64(ubar1, (nspec, numg, numt))
    tl, pl = f64(tlevel_3d, (nspec, nlevel, numg, numt)), f64(plevel_3d, (nspec, nlevel, numg, numt))
    gw = f64(gweight) if gweight is not None else None
    tw = f64(tweight) if tweight is not None else None
    a_dt, p_dt = _ptr_array(list(dtau_3d))
    a_w0, p_w0 = _ptr_array(list(w0_3d))
    a_cb, p_cb = _ptr_array(list(cosb_3d)) if cosb_3d is not None else (None, None)
    a_rs, p_rs = _ptr_array(_per_spectrum(surf_reflect, nspec))
    a_fx, p_fx = _ptr_array(_per_spectrum(int_at_top, nspec))
    fuse = flux_disk is not None and gw is not None and tw is not None
    a_fd, p_fd = _ptr_array(_per_spectrum(flux_disk, nspec)) if fuse else (None, None)
    check(load().picaso_get_thermal_3d_batch_dev(
        ctx, _ci(nspec), _ci(nlevel), _addr(wno), _ci(nwno), _ci(numg), _ci(numt), ptr(tl), p_dt, p_w0, p_cb, ptr(pl),
        ptr(u1), p_rs, _ci(int(hard_surface)), p_fx, ptr(gw) if fuse else None, ptr(tw) if fuse else None, p_fd), ctx)


# ------------------------------------------------------------------------------------------------
# the correlated-k Gauss-point loop around the SH and the 3-D solvers (csrc/ckloop.hip)
# ------------------------------------------------------------------------------------------------
def _opt(x):
    return ptr(x) if x is not None else None


def reflected_SH_ck(ctx, nlevel, nwno, ngauss, numg, numt, planes, surf_reflect, ubar0, ubar1, cos_theta, F0PI,
                    w_single_form, w_multi_form, psingle_form, w_single_rayleigh, w_multi_rayleigh, psingle_rayleigh,
                    frac_a, frac_b, frac_c, constant_back, constant_forward, stream, gauss_wts, xint_at_top, b_top=0.0,
                    single_form=0, compound_f_deltaM=True, cloud_free_above=0, gweight=None, tweight=None, albedo=None):
    """The reference's ``for ig in range(ngauss)`` loop around ``get_reflected_SH`` (justdoit.py:256-269, :307) as one
    launch: ``planes`` maps ``SH_PLANES`` to ``(nlayer|nlevel, nwno, ngauss)`` DeviceArrays as ``compute_opacity`` lays
    them out; ``xint_at_top`` the Gauss-weighted ``(numg, numt, nwno)`` result (``picaso_get_reflected_SH_ck_dev``)."""
    u0, u1 = f64(ubar0, (numg, numt)), f64(ubar1, (numg, numt))
    gw = f64(gweight) if gweight is not None else None
    tw = f64(tweight) if tweight is not None else None
    check(load().picaso_get_reflected_SH_ck_dev(
        ctx, _ci(nlevel), _ci(nwno), _ci(ngauss), _ci(numg), _ci(numt), *[_addr(planes.get(k)) for k in SH_PLANES],
        _addr(surf_reflect), ptr(u0), ptr(u1), _cd(cos_theta), _addr(F0PI), _ci(int(w_single_form)),
        _ci(int(w_multi_form)), _ci(int(psingle_form)), _ci(int(w_single_rayleigh)), _ci(int(w_multi_rayleigh)),
        _ci(int(psingle_rayleigh)), _cd(frac_a), _cd(frac_b), _cd(frac_c), _cd(constant_back), _cd(constant_forward),
        _ci(int(stream)), _cd(b_top), _ci(int(single_form)), _ci(1 if compound_f_deltaM else 0),
        _ci(int(cloud_free_above)), ptr(f64(gauss_wts, (ngauss,))), _addr(xint_at_top), _opt(gw), _opt(tw),
        _addr(albedo)), ctx)


def thermal_SH_ck(ctx, nlevel, wno, nwno, ngauss, numg, numt, tlevel, dtau, w0, cosb_og, plevel, ubar1, surf_reflect,
                  stream, hard_surface, cosb_differs_from_cosb_og, gauss_wts, xint_at_top, tau=None, gweight=None,
                  tweight=None, flux_disk=None):
    """The ``ngauss`` loop around ``get_thermal_SH`` (justdoit.py:364-370, :380) as one launch
    (``picaso_get_thermal_SH_ck_dev``): planes ``(nlayer, nwno, ngauss)``; ``cosb_differs_from_cosb_og`` is the
    reference's ``np.array_equal(cosb, cosb_og)`` test (fluxes.py:3072-3075), evaluated by the caller."""
    gw = f64(gweight) if gweight is not None else None
    tw = f64(tweight) if tweight is not None else None
    check(load().picaso_get_thermal_SH_ck_dev(
        ctx, _ci(nlevel), _addr(wno), _ci(nwno), _ci(ngauss), _ci(numg), _ci(numt), ptr(f64(tlevel, (nlevel,))),
        _addr(dtau), _addr(tau), _addr(w0), _addr(cosb_og), ptr(f64(plevel, (nlevel,))), ptr(f64(ubar1, (numg, numt))),
        _addr(surf_reflect), _ci(int(stream)), _ci(int(hard_surface)), _ci(1 if cosb_differs_from_cosb_og else 0),
        ptr(f64(gauss_wts, (ngauss,))), _addr(xint_at_top), _opt(gw), _opt(tw), _addr(flux_disk)), ctx)


def reflected_3d_ck(ctx, nlevel, nwno, ngauss, numg, numt, planes, surf_reflect, ubar0, ubar1, cos_theta, F0PI,
                    single_phase, multi_phase, frac_a, frac_b, frac_c, constant_back, constant_forward, gauss_wts,
                    xint_at_top, gweight=None, tweight=None, albedo=None):
    """The ``ngauss`` loop around ``get_reflected_3d`` (justdoit.py:488-500) as one launch on FACET-MAJOR planes
    ``(numg*numt, nlayer|nlevel, nwno, ngauss)`` (``optics.compute_opacity_facet_major_ck``); planes missing from
    ``planes`` are passed as NULL and re-derived (``picaso_get_reflected_3d_ck_dev``)."""
    gw = f64(gweight) if gweight is not None else None
    tw = f64(tweight) if tweight is not None else None
    check(load().picaso_get_reflected_3d_ck_dev(
        ctx, _ci(nlevel), _ci(nwno), _ci(ngauss), _ci(numg), _ci(numt), *[_addr(planes.get(k)) for k in REFLECTED_PLANES],
        _addr(surf_reflect), ptr(f64(ubar0, (numg, numt))), ptr(f64(ubar1, (numg, numt))), _cd(cos_theta), _addr(F0PI),
        _ci(single_phase), _ci(multi_phase), _cd(frac_a), _cd(frac_b), _cd(frac_c), _cd(constant_back),
        _cd(constant_forward), ptr(f64(gauss_wts, (ngauss,))), _addr(xint_at_top), _opt(gw), _opt(tw), _addr(albedo)), ctx)


def thermal_3d_ck(ctx, nlevel, wno, nwno, ngauss, numg, numt, tlevel_3d, dtau, w0, cosb, plevel_3d, ubar1, surf_reflect,
                  hard_surface, gauss_wts, int_at_top, gweight=None, tweight=None, flux_disk=None):
    """The ``ngauss`` loop around ``get_thermal_3d`` (justdoit.py:502-516) as one launch on facet-major planes
    ``(numg*numt, nlayer, nwno, ngauss)``; ``tlevel_3d`` / ``plevel_3d`` host ``(nlevel, numg, numt)``
    (``picaso_get_thermal_3d_ck_dev``)."""
    gw = f64(gweight) if gweight is not None else None
    tw = f64(tweight) if tweight is not None else None
    check(load().picaso_get_thermal_3d_ck_dev(
        ctx, _ci(nlevel), _addr(wno), _ci(nwno), _ci(ngauss), _ci(numg), _ci(numt),
        ptr(f64(tlevel_3d, (nlevel, numg, numt))), _addr(dtau), _addr(w0), _addr(cosb),
        ptr(f64(plevel_3d, (nlevel, numg, numt))), ptr(f64(ubar1, (numg, numt))), _addr(surf_reflect),
        _ci(int(hard_surface)), ptr(f64(gauss_wts, (ngauss,))), _addr(int_at_top), _opt(gw), _opt(tw),
        _addr(flux_disk)), ctx)
