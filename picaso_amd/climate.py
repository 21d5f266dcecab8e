"""Radiative-transfer call of the climate solver on the GPU: ``get_fluxes``.

Drop-in for the reference's ``climate.get_fluxes`` (picaso/climate.py:1687-1952), the function its
T(P) iteration evaluates on every Newton step: Toon reflected light at the two-stream angle
``ubar = 0.5`` and Toon thermal emission with bin-integrated Planck functions (``calc_type=1``),
both returning level and layer-midpoint fluxes for every correlated-k Gauss point, then the
Gauss-weight, patchy-cloud, disk and wavenumber sums.  Same positional arguments (the reference's
namedtuples or anything with the same attribute names) and the same eight arrays back.

Here the ``ngauss`` loop is one launch over ``nwno*ngauss`` columns per solver
(``picaso_get_reflected_1d_ck_dev`` / ``picaso_get_thermal_1d_ck_dev``, level-flux kernels of
``toon_lvl.hip``), the cloudy/clear blend and the disk quadrature run on the device, and only the
``(nlevel, nwno)`` results come back for the final wavenumber sums.
"""
from collections import namedtuple

import os

import numpy as np

from . import _lib, resident
from ._lib import f64
from .device import DeviceArray, PinnedArray

# the reference's containers (climate.py:1962-1966)
Atmosphere_Tuple = namedtuple("Atmosphere_Tuple", ["dtdp", "mmw_layer", "nlevel", "t_level", "p_level", "condensables",
                                                   "condensable_abundances", "condensable_weights", "scale_height"])
OpacityWEd_Tuple = namedtuple("OpacityWEd_Tuple", ["DTAU", "TAU", "W0", "COSB", "ftau_cld", "ftau_ray", "GCOS2",
                                                   "W0_no_raman", "f_deltaM"])
OpacityNoEd_Tuple = namedtuple("OpacityNoEd_Tuple", ["DTAU", "TAU", "W0", "COSB"])
ScatteringPhase_Tuple = namedtuple("ScatteringPhase_Tuple", ["surf_reflect", "single_phase", "multi_phase", "frac_a",
                                                             "frac_b", "frac_c", "constant_back", "constant_forward"])
Disco_Tuple = namedtuple("Disco_Tuple", ["ng", "nt", "gweight", "tweight", "ubar0", "ubar1", "cos_theta"])
Opagrid_Tuple = namedtuple("Opagrid_Tuple", ["nwno", "delta_wno", "wno", "ngauss", "gauss_wts"])


@_lib.serialized
def calculate_atm(bundle, opacityclass, only_atmosphere=False):
    """Atmosphere set-up and opacities of one climate iteration (reference ``climate.calculate_atm``,
    climate.py:1969-2135): returns ``OpacityWEd, OpacityNoEd, ScatteringPhase, Disco, Atmosphere,
    (OpacityWEd_hole, OpacityNoEd_hole)`` with the reference's namedtuples.  The opacity planes are
    HBM-resident ``DeviceArray`` objects ``(nlayer|nlevel, nwno, ngauss)`` -- ``get_fluxes`` takes them as
    they are (``.to_host()`` gives the reference's numpy arrays)."""
    from . import justdoit, optics
    inputs = bundle.inputs
    opa = opacityclass
    common = inputs["approx"]["rt_params"]["common"]
    toon = inputs["approx"]["rt_params"]["toon"]
    frac_a, frac_b, frac_c = common["TTHG_params"]["fraction"]
    geom = inputs["disco"]
    do_holes = bool(inputs["clouds"].get("do_holes", False))
    atm = justdoit._setup_atmosphere(inputs, opa, opa.wno)
    atm.surf_reflect = 0                                   # climate.py:2052
    atm.get_dtdp()
    prof = inputs["atmosphere"]["profile"]
    ours = [m for m in ("H2O", "CH4", "NH3", "Fe") if m in prof.keys()]                  # :2090-2093
    Atmosphere = Atmosphere_Tuple(atm.layer["dtdp"], atm.layer["mmw"], atm.c.nlevel,
                                  np.ascontiguousarray(atm.level["temperature"]).copy(),
                                  np.ascontiguousarray(atm.level["pressure_bar"]).copy(), ours,
                                  np.array([np.asarray(prof[m], dtype=float) for m in ours]),
                                  [atm.weights[m] for m in ours], atm.level["scale_height"])
    if only_atmosphere:
        return Atmosphere
    opa.get_opacities(atm)
    kw = dict(ngauss=opa.ngauss, stream=common["stream"], delta_eddington=common["delta_eddington"],
              test_mode=inputs["test_mode"], raman=common["raman"])

    def tuples(pl):
        return (OpacityWEd_Tuple(pl["dtau"], pl["tau"], pl["w0"], pl["cosb"], pl["ftau_cld"], pl["ftau_ray"],
                                 pl["gcos2"], pl["w0_no_raman"], pl["f_deltaM"]),
                OpacityNoEd_Tuple(pl["dtau_og"], pl["tau_og"], pl["w0_og"], pl["cosb_og"]))
    holes = (None, None)
    if do_holes:                                           # :2104-2112
        holes = tuples(optics.compute_opacity_resident(atm, opa, fthin_cld=inputs["clouds"]["fthin_cld"],
                                                       do_holes=True, **kw))
    wed, noed = tuples(optics.compute_opacity_resident(atm, opa, **kw))
    sp = ScatteringPhase_Tuple(atm.surf_reflect, toon["single_phase"], toon["multi_phase"], frac_a, frac_b, frac_c,
                               common["TTHG_params"]["constant_back"], common["TTHG_params"]["constant_forward"])
    dis = Disco_Tuple(geom["num_gangle"], geom["num_tangle"], geom["gweight"], geom["tweight"], geom["ubar0"],
                      geom["ubar1"], geom["cos_theta"])
    return wed, noed, sp, dis, Atmosphere, holes


def _planes(wed, noed, ctx, thermal_only=False):
    """Upload the (rows, nwno, ngauss) arrays of one opacity set (DeviceArrays pass through)."""
    def up(x):
        return x if isinstance(x, DeviceArray) else DeviceArray.from_host(f64(x), ctx)
    pl = {"dtau_og": up(noed.DTAU), "w0_no_raman": up(wed.W0_no_raman), "cosb_og": up(noed.COSB)}
    if not thermal_only:
        pl.update(dtau=up(wed.DTAU), tau=up(wed.TAU), w0=up(wed.W0), cosb=up(wed.COSB), gcos2=up(wed.GCOS2),
                  ftau_cld=up(wed.ftau_cld), ftau_ray=up(wed.ftau_ray), tau_og=up(noed.TAU), w0_og=up(noed.W0))
    return pl


_VEC_CACHE = {}          # (pid, context, bytes) -> DeviceArray; entries of a context go when it is destroyed
_lib.on_context_destroy(lambda value: [_VEC_CACHE.pop(k) for k in [k for k in _VEC_CACHE if k[1] == value]])


def _resident_small(values, ctx):
    """A per-wavelength host vector as a resident one, kept by content: the T(P) iteration calls get_fluxes thousands of
    times with the same wavenumbers, bin widths, stellar flux and surface reflectivity (four 5 KB uploads per call,
    0.06 ms each)."""
    a = np.ascontiguousarray(values, dtype=np.float64)
    key = (os.getpid(), getattr(ctx, "value", ctx), a.tobytes())
    hit = _VEC_CACHE.get(key)
    if hit is None:
        while len(_VEC_CACHE) >= 64:               # oldest first
            del _VEC_CACHE[next(iter(_VEC_CACHE))]
        hit = _VEC_CACHE[key] = DeviceArray.from_host(a, ctx)
    return hit


@_lib.serialized
def get_fluxes(Atmosphere, OpacityWEd, OpacityNoEd, ScatteringPhase, Disco, Opagrid, F0PI, reflected, thermal,
               do_holes=False, fhole=0.0, hole_OpacityWEd=None, hole_OpacityNoEd=None, ctx=None,
               copy_outputs=False):
    """Visible and IR net (layer and level), upward and downward fluxes (reference
    ``climate.get_fluxes``, climate.py:1687-1952).  Returns ``flux_net_v_layer, flux_net_v, flux_plus_v,
    flux_minus_v, flux_net_ir_layer, flux_net_ir, flux_plus_ir, flux_minus_ir``.

    ``flux_plus_v`` / ``flux_minus_v`` are ``(ng, nt, nlevel, nwno)`` as in the reference, where every
    disk angle holds the same two-stream result (climate.py:1803-1805, :1868-1869); here they are
    read-only broadcast views of that one ``(nlevel, nwno)`` array (the climate solver reads
    ``[0, 0, :, :]``, climate.py:946-947) unless ``copy_outputs=True`` asks for writable copies."""
    ctx = ctx if ctx is not None else _lib.context()
    pressure, temperature, nlevel = Atmosphere.p_level, Atmosphere.t_level, int(Atmosphere.nlevel)
    sp = ScatteringPhase
    ng, nt = int(Disco.ng), int(Disco.nt)
    nwno, ngauss = int(Opagrid.nwno), int(Opagrid.ngauss)
    dwni, wno, gauss_wts = f64(Opagrid.delta_wno), f64(Opagrid.wno), f64(Opagrid.gauss_wts)
    if do_holes and (hole_OpacityWEd is None or hole_OpacityNoEd is None):
        raise Exception("get_fluxes: do_holes=True needs hole_OpacityWEd and hole_OpacityNoEd")

    flux_net_v = np.zeros((ng, nt, nlevel))
    flux_net_v_layer = np.zeros((ng, nt, nlevel))
    flux_plus_v = flux_minus_v = None
    flux_net_ir = np.zeros(nlevel)
    flux_net_ir_layer = np.zeros(nlevel)
    flux_plus_ir = np.zeros((nlevel, nwno))
    flux_minus_ir = np.zeros((nlevel, nwno))

    rs = _resident_small(np.zeros(nwno) + f64(sp.surf_reflect), ctx)
    if thermal:                                           # uploaded (first use) before the second stream is ordered behind this one
        d_wno, d_dw = _resident_small(wno, ctx), _resident_small(dwni, ctx)
    sets = [_planes(OpacityWEd, OpacityNoEd, ctx, thermal_only=not reflected)]
    if do_holes:
        sets.append(_planes(hole_OpacityWEd, hole_OpacityNoEd, ctx, thermal_only=not reflected))

    def blend(results, c):                                # (1-fhole)*cloudy + fhole*clear, climate.py:1838-1842
        if len(results) == 1:
            return results[0]
        for a, b in zip(*results):
            resident.axpby(c, 1.0 - fhole, a, fhole, b, a)
        return results[0]

    # both legs are small launches (661 bins x 8 Gauss points = 83 waves on a 1 024-SIMD chip, one 90-layer chain each):
    # the thermal leg goes to the process's second stream, behind the uploads above, and runs next to the reflected one
    tctx = ctx
    if reflected and thermal and os.environ.get("PICASO_AMD_OVERLAP_LEGS", "1") != "0":
        tctx = _lib.aux_context(_lib.device_of(ctx))
        _lib.ctx_wait(tctx, ctx)

    if reflected:                                         # climate.py:1796-1874
        d_f0 = _resident_small(np.zeros(nwno) + f64(F0PI), ctx)
        half = np.full((1, 1), 0.5)                       # ubar0_clima = ubar1_clima = 0.5, one angle
        xdummy = DeviceArray((1, 1, nwno), ctx)
        res = []
        for pl in sets:
            stack = DeviceArray((4, 1, 1, nlevel, nwno), ctx)       # one buffer, one copy back
            lv = [stack.row_block(k) for k in range(4)]
            resident.reflected_1d_ck(ctx, nlevel, nwno, ngauss, 1, 1, pl, rs, half, half, float(Disco.cos_theta),
                                     d_f0, int(sp.single_phase), int(sp.multi_phase), float(sp.frac_a),
                                     float(sp.frac_b), float(sp.frac_c), float(sp.constant_back),
                                     float(sp.constant_forward), gauss_wts, xdummy, get_toa_intensity=0,
                                     lvl_fluxes=lv)
            res.append(lv)
        refl_stack = blend(res, ctx)[0]._owner            # read back after the thermal leg is enqueued
        refl_pin = refl_stack.to_host_async(PinnedArray(refl_stack.shape, ctx))     # the copy: behind the leg's last kernel

    if thermal:                                           # climate.py:1879-1941
        xdummy = DeviceArray((ng, nt, nwno), tctx)
        res = []
        for pl in sets:
            lv = [DeviceArray((ng, nt, nlevel, nwno), tctx) for _ in range(4)]
            resident.thermal_1d_ck(tctx, nlevel, d_wno, nwno, ngauss, ng, nt, temperature, pl["dtau_og"],
                                   pl["w0_no_raman"], pl["cosb_og"], pressure, Disco.ubar1, rs, 0, gauss_wts,
                                   xdummy, dwno=d_dw, calc_type=1, lvl_fluxes=lv)
            res.append(lv)
        disk = DeviceArray((4, nlevel, nwno), tctx)
        for k, x in enumerate(blend(res, tctx)):          # compress_thermal over the disk angles (:1925-1928)
            resident.compress_thermal(tctx, nlevel * nwno, x, Disco.gweight, Disco.tweight, disk.row_block(k))
        therm_pin = disk.to_host_async(PinnedArray(disk.shape, tctx))

    # both legs are on the stream before the first copy back (each copy is a synchronisation)
    if reflected:
        fm, fp, fmm, fpm = refl_pin.wait().copy()         # Gauss-weighted (1,1,nlevel,nwno) each
        refl_pin.free()
        flux_net_v_layer += np.sum(fpm, axis=3) - np.sum(fmm, axis=3)
        flux_net_v += np.sum(fp, axis=3) - np.sum(fm, axis=3)
        # the single two-stream angle stands for every disk angle (climate.py:1803-1805, :1868-1869)
        flux_plus_v = np.broadcast_to(fp, (ng, nt, nlevel, nwno))
        flux_minus_v = np.broadcast_to(fm, (ng, nt, nlevel, nwno))
        if copy_outputs:
            flux_plus_v, flux_minus_v = flux_plus_v.copy(), flux_minus_v.copy()
    if thermal:
        fm, fp, fmm, fpm = therm_pin.wait()
        flux_net_ir_layer = ((fpm - fmm) * dwni).sum(axis=1)                  # (:1931-1936)
        flux_net_ir = ((fp - fm) * dwni).sum(axis=1)
        flux_plus_ir = fp * dwni
        flux_minus_ir = fm * dwni
        therm_pin.free()

    if flux_plus_v is None:
        flux_plus_v, flux_minus_v = np.zeros((ng, nt, nlevel, nwno)), np.zeros((ng, nt, nlevel, nwno))
    return (flux_net_v_layer, flux_net_v, flux_plus_v, flux_minus_v, flux_net_ir_layer, flux_net_ir,
            flux_plus_ir, flux_minus_ir)


@_lib.serialized
def get_fluxes_tbatch(temperatures, Atmosphere, OpacityWEd, OpacityNoEd, ScatteringPhase, Disco, Opagrid, ctx=None,
                      chunk=32, nets_only=False):
    """The IR half of ``get_fluxes`` (``reflected=False, thermal=True``) for every level-temperature profile in
    ``temperatures`` ``(nitem, nlevel)`` over ONE set of opacities: ``flux_net_ir_layer, flux_net_ir`` ``(nitem, nlevel)``
    and ``flux_plus_ir, flux_minus_ir`` ``(nitem, nlevel, nwno)``.

    What it is for: the Jacobian of the reference's T(P) iteration perturbs one level temperature at a time, rebuilds
    the profile and calls ``get_fluxes`` with the SAME opacities (climate.py:1105-1180: ~nlevel calls per Newton step,
    each a Planck evaluation + two-stream solve + source-function sweeps).  Here the profiles are extra columns of one
    launch sequence (``picaso_get_thermal_1d_ck_tbatch_dev``: a column reads the shared planes and its own profile's
    level temperatures); ``chunk`` profiles at a time bound the scratch (91 levels x 661 bins x 8 Gauss points x 5 angles:
    77 MB per profile).  Row ``k`` of every result equals ``get_fluxes(Atmosphere._replace(t_level=temperatures[k]), ...)``
    bit for bit (same kernels per column, same numpy sums).  Patchy clouds (``do_holes``) blend the cloudy and the clear
    column sets before the disk sum: call ``get_fluxes`` per profile for those.

    ``nets_only=True`` (what the Jacobian reads, climate.py:1182-1185): only ``flux_net_ir_layer, flux_net_ir`` come back,
    summed over wavenumber ON THE DEVICE (``picaso_flux_net_sums_dev``: a fixed tree per row instead of numpy's order, so
    they agree with ``get_fluxes`` to ~1e-15 relative instead of bit for bit) -- 2 x nlevel doubles per profile cross
    PCIe instead of 4 x nlevel x nwno."""
    import ctypes
    ctx = ctx if ctx is not None else _lib.context()
    temps = f64(temperatures)
    nlevel = int(Atmosphere.nlevel)
    if temps.ndim != 2 or temps.shape[1] != nlevel:
        raise Exception("get_fluxes_tbatch: temperatures must be (nitem, nlevel=%d)" % nlevel)
    nitem = temps.shape[0]
    sp = ScatteringPhase
    ng, nt = int(Disco.ng), int(Disco.nt)
    nwno, ngauss = int(Opagrid.nwno), int(Opagrid.ngauss)
    dwni, wno, gauss_wts = f64(Opagrid.delta_wno), f64(Opagrid.wno), f64(Opagrid.gauss_wts)
    rs = _resident_small(np.zeros(nwno) + f64(sp.surf_reflect), ctx)
    pl = _planes(OpacityWEd, OpacityNoEd, ctx, thermal_only=True)
    d_wno, d_dw = _resident_small(wno, ctx), _resident_small(dwni, ctx)
    net_layer, net = np.empty((nitem, nlevel)), np.empty((nitem, nlevel))
    plus = minus = None
    if not nets_only:
        plus, minus = np.empty((nitem, nlevel, nwno)), np.empty((nitem, nlevel, nwno))
    for c0 in range(0, nitem, max(1, int(chunk))):
        tl = temps[c0:c0 + max(1, int(chunk))]
        m = tl.shape[0]
        disk4 = DeviceArray((4, nlevel, m * nwno), ctx)
        resident.thermal_1d_ck_tbatch(ctx, nlevel, d_wno, nwno, ngauss, ng, nt, tl, pl["dtau_og"], pl["w0_no_raman"],
                                      pl["cosb_og"], Atmosphere.p_level, Disco.ubar1, rs, 0, gauss_wts, Disco.gweight,
                                      Disco.tweight, disk4, dwno=d_dw, calc_type=1)
        if nets_only:
            d_nl, d_n = DeviceArray((m, nlevel), ctx), DeviceArray((m, nlevel), ctx)
            _lib.check(_lib.load().picaso_flux_net_sums_dev(ctx, ctypes.c_int(nlevel), ctypes.c_int(m), ctypes.c_int(nwno),
                                                            ctypes.c_void_p(disk4.addr), ctypes.c_void_p(d_dw.addr),
                                                            ctypes.c_void_p(d_nl.addr), ctypes.c_void_p(d_n.addr)), ctx)
            net_layer[c0:c0 + m], net[c0:c0 + m] = d_nl.to_host(), d_n.to_host()
            continue
        fm, fp, fmm, fpm = disk4.to_host().reshape(4, nlevel, m, nwno)
        for k in range(m):                                     # get_fluxes' own expressions (climate.py:1931-1936)
            net_layer[c0 + k] = ((fpm[:, k] - fmm[:, k]) * dwni).sum(axis=1)
            net[c0 + k] = ((fp[:, k] - fm[:, k]) * dwni).sum(axis=1)
            plus[c0 + k] = fp[:, k] * dwni
            minus[c0 + k] = fm[:, k] * dwni
    if nets_only:
        return net_layer, net
    return net_layer, net, plus, minus
