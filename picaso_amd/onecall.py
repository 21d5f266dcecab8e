"""One C call per spectrum (csrc/driver.hip): the launch sequence of the 1-D path (Toon or SH) or of the 3-D path for every
wavelength block.

The plain 1-D spectrum (reference justdoit.py:236-385, 552-599) through ``picaso_toon_spectrum_blocks``: ONE C call
enqueues gas stage -> ``compute_opacity`` -> reflected || thermal (+ fused disk sums) on every wavelength block of ``subs``
(``[(lo, hi, opacity object of the block)]``; the whole grid on one GPU is one block), a second and third copy the legs
back.  Outside what the driver covers (correlated-k tables, SH layer fluxes, patchy clouds, level fluxes, full_output,
transmission, Oklopcic Raman in a multi-block call, test modes) ``prepare`` returns None and the caller
takes ``spectrum.Spectrum``, whose results these are bit for bit: the C function chains the same entry points in the same
order.  ``prepare`` does everything up to the C call -- set-up, block table, per-call pointers, job -- ``run`` is
prepare + the call + ``finish``.
"""
import ctypes

import numpy as np

from . import _lib, device, optics, resident
from . import driver as drv
from .atmsetup import CloudTables
from .device import DeviceArray
from .spectrum import (_bond_denominator, _constant_planes, _ones, _post_final, _post_reflected, _post_thermal,
                       _resident_vector, _setup_atmosphere, _trapz_resident)


def run(bundle, opa, subs, calculation, opt, dimension="1d"):
    """``prepare`` + the C call + ``finish``; None when the call is outside what the driver covers."""
    p = prepare_3d(bundle, opa, subs, calculation, opt) if dimension == "3d" else \
        prepare(bundle, opa, subs, calculation, opt, early=True)
    if p is None:
        return None
    try:
        drv.enqueue(p["table"], p["job"], p.get("phase", 0))
        return finish(p)
    except BaseException:
        drv.abandon(p["table"])
        raise


def _in_scope(inp, opa, legs, nblocks, opt):
    """The calls the C driver covers: reflected and / or thermal, Toon or SH (without layer fluxes), monochromatic
    resident tables, no patchy clouds (SH ignores them, as the reference: ``Spectrum`` says so with a warning and is left to
    say it), no level fluxes, no test mode; Oklopcic's Raman plane (formed per call on the block's device) for one block
    only."""
    if opt.no_driver or opt.raman_planes or opt.unfused_opacity:
        return False                # (unfused_opacity: the C driver always takes the fused launch; Spectrum honours the option)
    if not legs or not legs <= {"reflected", "thermal"}:
        return False
    is_sh = inp["approx"]["rt_method"] == "SH"
    if opa.ngauss != 1 and (is_sh or getattr(opa, "_kappa", None) is None or inp["approx"].get("get_lvl_flux", False)
                            or inp["approx"]["rt_params"]["common"]["raman"] == 0):
        return False                # k-tables: premixed, Toon, TOA intensities (the SH / level-flux forms: Spectrum)
    if (getattr(opa, "on_fly", False) or inp["clouds"].get("do_holes", False)
            or (inp["approx"].get("get_lvl_flux", False) and not is_sh)
            or inp["test_mode"] is not None or not hasattr(opa, "_cia") or not hasattr(opa, "_ray")):
        return False
    if is_sh and inp["approx"]["rt_params"]["SH"]["calculate_fluxes"]:
        return False                # the layer moment fluxes (flx = 1) are a per-call output of the call-by-call path
    return not (inp["approx"]["rt_params"]["common"]["raman"] == 0 and nblocks != 1)


_SH_READS = ("dtau", "w0", "cosb_og", "ftau_cld", "ftau_ray", "f_deltaM", "dtau_og", "w0_og")


def _plane_set_sh(inp, atm, nwno, common, frac_c, opt):
    """``_plane_set`` for the SH solvers: ``(want, lean, sh_top)`` as ``Spectrum._want_1d`` chooses them -- dtau and w0
    for a cloud-free atmosphere with the default options, the eight planes the launch reads when the level planes may be
    derived, all thirteen otherwise; ``sh_top`` = the cloud-free layers above the deck."""
    from .spectrum import _cloud_free_top
    sh = inp["approx"]["rt_params"]["SH"]
    forms = (sh["w_single_form"], sh["w_multi_form"], sh["psingle_form"], sh["w_single_rayleigh"], sh["w_multi_rayleigh"],
             sh["psingle_rayleigh"], frac_c, sh["single_form"], 0)
    rayleigh = len(getattr(atm, "rayleigh_molecules", [])) > 0
    lean = (bool(getattr(atm, "cloud_free", False)) and rayleigh and not opt.all_planes
            and resident.reflected_SH_can_derive(common["stream"], *forms))
    if lean:
        return {"dtau", "w0"}, True, 0
    sh_top = _cloud_free_top(inp, atm.c.nlayer) if (rayleigh and not opt.all_planes) else 0
    if not opt.all_planes and resident.reflected_SH_can_derive_levels(atm.c.nlevel, nwno, common["stream"], *forms):
        return set(_SH_READS), False, sh_top
    return set(drv.OUT_NAMES), False, sh_top


def _plane_set(atm, geom, toon, frac_c, nwno, raman, do_r, do_t, opt):
    """Which planes compute_opacity writes: exactly ``Spectrum._want_1d``'s choice (see there for the cloud-free form and
    for the planes the reflected kernels re-derive).  Returns ``(want, lean, derive)``."""
    ng, nt = geom["num_gangle"], geom["num_tangle"]
    derive = (do_r and not opt.all_planes
              and resident.reflected_can_derive(atm.c.nlevel, nwno, ng, nt, geom["ubar0"], geom["ubar1"], geom["cos_theta"],
                                                toon["single_phase"], toon["multi_phase"], frac_c,
                                                toon["toon_coefficients"], False))
    lean = (bool(getattr(atm, "cloud_free", False)) and len(getattr(atm, "rayleigh_molecules", [])) > 0
            and not opt.all_planes)
    want = set()
    if lean:
        if do_r:
            want |= {"dtau", "w0"} if derive else {"dtau", "tau", "w0"}
        if do_t:
            want |= {"dtau", "w0" if (raman == 2 and do_r) else "w0_no_raman"}
        return want, lean, derive
    if do_r:
        want |= set(resident.REFLECTED_PLANES)
        if derive:
            want -= {"tau", "tau_og", "gcos2"}
    if do_t:
        want |= {"dtau_og", "w0_no_raman", "cosb_og"}
    return want, lean, derive


def _cloud_inputs(atm, opa, tables, nlayer, nwno, opt, hold):
    """The cloud tables of the call as ``(device planes | None, device tables on their own grid | None, host planes |
    None)``: tables on their own wavenumber grid (what virga and the box-cloud form of clouds() hand over) are regridded on
    the device as in ``compute_opacity_resident`` -- interpolated inside the opacity launch, no regridded planes in HBM
    (same bits) -- arrays already on the opacity grid travel as host planes, block by block."""
    cld = atm.layer["cloud"]
    if tables:
        stack = cld.__dict__.get("_stack")
        if stack is None:
            stack = cld.__dict__["_stack"] = np.concatenate([cld.compact[k] for k in ("opd", "w0", "g0")])
        if opt.unfused_opacity or opt.regrid_planes:
            all3 = device.regrid_rows(cld.in_wno, stack, optics._wno_device(opa, cld.wno), opa.ctx).reshape((3, nlayer, nwno))
            dcld = [all3.row_block(0), all3.row_block(1), all3.row_block(2)]
            hold.append((all3, dcld))
            return dcld, None, None
        # the compact tables go to every block's device (0.4 MB): each block interpolates them to its own wavenumbers
        xp, fp = np.ascontiguousarray(cld.in_wno, dtype=np.float64), np.ascontiguousarray(stack, dtype=np.float64)
        dtab = (int(np.size(cld.in_wno)), xp, fp, optics.content_digest(xp) + optics.content_digest(fp))
        return None, dtab, None
    if getattr(atm, "cloud_free", False):
        return None, None, None

    def plane(x):
        a = np.asarray(x, dtype=float)
        return a if (a.shape == (nlayer, nwno) and a.flags.c_contiguous) else \
            np.ascontiguousarray(np.broadcast_to(a, (nlayer, nwno)))
    hcld = [plane(cld[k]) for k in ("opd", "w0", "g0")]
    hold.append(hcld)
    return None, None, hcld


def _call_state(inp, opa, subs, opt, atm, raman, do_r, do_t, table, ng, nt):
    """What ``_fill_block`` needs of one call, for 1-D and 3-D alike: stellar inputs, the full-grid result arrays (one
    element longer when the spectrum-wide integrals arrive with them), the list that keeps per-call device objects alive."""
    nwno = opa.nwno
    nostar = inp["star"]["database"] == "nostar"
    F0PI = _ones(opa, nwno) if nostar else inp["star"]["relative_flux"]
    stellar = getattr(opa, "unshifted_stellar_spec", None)
    if stellar is None:
        stellar = F0PI
    integrals = len(subs) == 1 and nwno > 1 and not opt.host_integrals
    full = {}
    if do_r:
        full["albedo"] = np.empty(nwno + 1 if integrals else nwno)
    if do_t:
        full["thermal"] = np.empty(nwno + 1 if integrals else nwno)
    return dict(inp=inp, opa=opa, nwno=nwno, wno=opa.wno, hold=[], atm=atm, nostar=nostar, F0PI=F0PI, stellar=stellar,
                nblocks=len(subs), raman=raman, clouds=(None, None, None), do_r=do_r, do_t=do_t,
                overlap=do_r and do_t and opt.overlap_legs, seen_dev={}, table=table, ng=ng, nt=nt, full=full,
                integrals=integrals, denom=None)


def _prepared(c, job, keep, signature):
    """The dictionary ``prepare`` / ``prepare_3d`` return (``finish`` reads it; ``keep`` holds what the job points into)."""
    return dict(table=c["table"], job=job, keep=(keep, c["hold"]), do_r=c["do_r"], do_t=c["do_t"], full=c["full"],
                nwno=c["nwno"], integrals=c["integrals"], denom=c["denom"] if (c["integrals"] and c["do_r"]) else None,
                wno=c["wno"], stellar=c["stellar"], inp=c["inp"], atm=c["atm"], opa=c["opa"], signature=signature)


def _fill_block(k, sub, lo, hi, c):
    """The per-call pointers of one wavelength block ``k`` (a ``driver.Block``): what the opacity stage reads
    (``_fill_block_opacity``), then what the legs read (``_fill_block_legs``)."""
    _fill_block_opacity(k, sub, lo, hi, c)
    _fill_block_legs(k, sub, lo, hi, c)


def _fill_block_opacity(k, sub, lo, hi, c):
    """The opacity stage's half: the Raman factor, the cloud inputs, and the context of the thermal leg (the stage orders
    that stream behind its gas launch)."""
    nw, nwno, hold, atm = hi - lo, c["nwno"], c["hold"], c["atm"]
    if c["raman"] == 1:
        row, _ = optics.raman_device(atm, sub, 1)
        k.raman = drv._dev(row)
    elif c["raman"] == 0:           # (nlayer, nwno) plane from the layer temperatures (picaso_raman_oklopcic_dev), same stream
        rplane, _ = optics.raman_device(atm, sub, 0)
        hold.append(rplane)
        k.raman = drv._dev(rplane)
    else:
        k.raman = None
    dcld, dtab, hcld = c["clouds"]
    k.cld_tab_nin, k.cld_tab_xp, k.cld_tab_fp = 0, None, None
    k.cld_opd = k.cld_w0 = k.cld_g0 = None
    k.cld_host_opd = k.cld_host_w0 = k.cld_host_g0 = None
    if dtab is not None:
        # kept on the block's opacity object while the tables' content is the same (a digest of every byte): a retrieval
        # that varies the gas keeps its cloud, and eight blocks otherwise pay sixteen uploads per call
        hit = sub.__dict__.get("_cloud_tables_dev")
        if hit is None or hit[0] != dtab[3]:
            hit = sub.__dict__["_cloud_tables_dev"] = (dtab[3], DeviceArray.from_host(dtab[1], sub.ctx),
                                                       DeviceArray.from_host(dtab[2], sub.ctx))
        d_xp, d_fp = hit[1], hit[2]
        hold.append((d_xp, d_fp))
        k.cld_tab_nin, k.cld_tab_xp, k.cld_tab_fp = dtab[0], drv._dev(d_xp), drv._dev(d_fp)
        k.wno = drv._dev(_resident_vector(sub, "wno", sub.wno, nw))
    elif dcld is not None:
        k.cld_opd, k.cld_w0, k.cld_g0 = (drv._dev(x) for x in dcld)
    elif hcld is not None:
        k.cld_host_opd, k.cld_host_w0, k.cld_host_g0 = (drv._host(h) for h in hcld)
        k.cld_host_pitch = nwno
    if c["do_t"]:
        tctx = sub.ctx
        if c["overlap"]:
            dev = _lib.device_of(sub.ctx)
            seen = c["seen_dev"]
            tctx = _lib.aux_context(dev, seen.get(dev, 0))       # blocks that share a device: a stream each
            seen[dev] = seen.get(dev, 0) + 1
        k.tctx = tctx.value if hasattr(tctx, "value") else tctx
        c.setdefault("tctx", {})[c["b"]] = tctx


def _fill_block_legs(k, sub, lo, hi, c):
    """The legs' half: resident per-wavelength vectors, the thermal workspace on the block's second stream, where the
    results go, the spectrum-wide integrals."""
    nw, nwno, hold, atm = hi - lo, c["nwno"], c["hold"], c["atm"]
    sr = atm.surf_reflect
    sr_full = np.ndim(sr) > 0 and np.size(sr) == nwno and nwno > 1
    rs = _resident_vector(sub, "surf_reflect", np.asarray(sr, dtype=float).reshape(nwno)[lo:hi] if sr_full else sr, nw)
    f0 = _resident_vector(sub, "F0PI", 1.0 if c["nostar"] else (c["F0PI"] if c["nblocks"] == 1 else c["F0PI"][lo:hi]), nw)
    k.surf_reflect, k.F0PI = drv._dev(rs), drv._dev(f0)
    hold.append((rs, f0))            # the block holds raw addresses: the vectors live as long as the call is in flight
    if c["do_t"]:
        tctx = c["tctx"][c["b"]]
        fl, dk, pin = c["table"].thermal_workspace(c["b"], tctx, c["ng"], c["nt"])
        k.flux, k.disk = drv._dev(fl), drv._dev(dk)
        k.thermal_pin = ctypes.cast(ctypes.c_void_p(pin.addr), drv._dp)
        k.wno = drv._dev(_resident_vector(sub, "wno", sub.wno, nw))
        k.thermal_host = drv._host(c["full"]["thermal"])
    if c["do_r"]:
        k.albedo_host = drv._host(c["full"]["albedo"])
    k.trapz_d = k.trapz_dr = k.stellar = None
    if c["integrals"]:
        # one block over the grid: the spectrum-wide integrals are formed on the device behind each result vector and
        # arrive with it (host arrays one element longer; the dictionary gets views of the first nwno)
        d_w, d_wr = _trapz_resident(sub, c["wno"])
        if c["do_r"]:
            d_st = f0 if c["stellar"] is c["F0PI"] else _resident_vector(sub, "stellar", c["stellar"], nw)
            hold.append(d_st)
            k.trapz_d, k.stellar = drv._dev(d_w), drv._dev(d_st)
            c["denom"] = _bond_denominator(sub, c["wno"], c["stellar"], d_st)
        if c["do_t"]:
            k.trapz_dr = drv._dev(d_wr)


def prepare(bundle, opa, subs, calculation, opt, slot=None, early=False):
    """Everything up to the C call: set-up, block table (``slot``: which of several tables of the same signature, for
    spectra that are in flight together), per-call pointers, job.  None: outside the driver's scope.  ``early=True`` (what
    ``run`` and ``picaso_async`` pass): the opacity stage is ALREADY on the stream when this returns and the returned
    dictionary says ``phase = 2`` -- the caller owes ``drv.enqueue(table, job, phase=2)``; with the default it says 0
    and nothing has been enqueued."""
    inp = bundle.inputs
    legs = set(calculation.split("+"))
    if not _in_scope(inp, opa, legs, len(subs), opt):
        return None
    common = inp["approx"]["rt_params"]["common"]
    toon = inp["approx"]["rt_params"]["toon"]
    raman = common["raman"]
    wno, nwno = opa.wno, opa.nwno
    atm = _setup_atmosphere(inp, opa, wno)
    cld = atm.layer["cloud"]
    cloud_free = bool(getattr(atm, "cloud_free", False))
    tables = not cloud_free and isinstance(cld, CloudTables)
    if tables and (np.size(cld.wno) != nwno or opt.host_regrid or opa.ngauss != 1 or
                   (len(subs) != 1 and (opt.unfused_opacity or opt.regrid_planes))):
        return None                 # tables on their own grid: interpolated inside each block's fused opacity launch
    nlevel, nlayer = atm.c.nlevel, atm.c.nlayer
    opa.get_opacities(atm, exclude_mol=inp["atmosphere"]["exclude_mol"])
    plan = opa._plan
    ck = bool(plan.get("premixed"))
    if ck and ("table" in plan or opa.ngauss < 2):
        return None                 # mixed on the fly: a per-call table (Spectrum)
    factors = optics._layer_factors(atm, opa)
    plan["_factors"] = (atm.layer["mixingratios"], factors)
    linear = opa.query_method == "linear"
    do_r, do_t = "reflected" in legs, "thermal" in legs
    geom = inp["disco"]
    ng, nt = geom["num_gangle"], geom["num_tangle"]
    frac_a, frac_b, frac_c = common["TTHG_params"]["fraction"]
    is_sh = inp["approx"]["rt_method"] == "SH"
    sh_top = 0
    if is_sh:
        want, lean, sh_top = _plane_set_sh(inp, atm, nwno, common, frac_c, opt)
        derive = False
    elif ck:                        # k-tables: the full set, as Spectrum._want_1d (no aliases, nothing derived)
        want, lean, derive = set(), False, False
        if do_r:
            want |= set(resident.REFLECTED_PLANES)
        if do_t:
            want |= {"dtau_og", "w0_no_raman", "cosb_og"}
    else:
        want, lean, derive = _plane_set(atm, geom, toon, frac_c, nwno, raman, do_r, do_t, opt)

    def table_ids(sub):          # a block table holds raw table addresses: replaced tables are a new signature
        if ck:
            return (id(sub._kappa),) + tuple(id(sub._cia[p]) for p in plan["cia_pairs"])
        mt = sub._mol_log if linear else sub._mol_raw
        return tuple(id(mt[m]) for m in plan["molecules"]) + tuple(id(sub._cia[p]) for p in plan["cia_pairs"])
    key = (tuple((lo, hi, id(sub)) + table_ids(sub) for lo, hi, sub in subs), nlayer, ng, nt, tuple(plan["molecules"]),
           tuple(plan["cia_pairs"]), tuple(factors[2]), linear, tuple(sorted(want)), lean, not cloud_free and not tables, do_r, do_t,
           derive, is_sh, opa.ngauss, slot)
    cache = opa.__dict__.setdefault("_driver_tables", {})
    table = cache.get(key)
    if table is None:
        if len(cache) > (8 if slot is None else 40):
            cache.clear()
        table = cache[key] = drv.BlockTable(subs, nlayer, ng, nt, plan["molecules"], plan["cia_pairs"], factors[2], linear,
                                            want, lean, not cloud_free and not tables, do_r, do_t, _constant_planes, derive,
                                            sh=is_sh, ngauss=opa.ngauss)
    c = _call_state(inp, opa, subs, opt, atm, raman, do_r, do_t, table, ng, nt)
    c["clouds"] = _cloud_inputs(atm, opa, tables, nlayer, nwno, opt, c["hold"])
    # The opacity stage first (round 6): as soon as the table rows / weights / coefficients are in the job and the cloud
    # inputs in the blocks, the gas kernel goes on the stream (drv.enqueue(phase=1)); geometry, level tables, resident
    # vectors, result buffers -- what the legs read -- are filled while it runs, and ``run`` enqueues the legs (phase 2).
    # The same launches in the same order: same bits (PICASO_AMD_ONE_PHASE=1 / Options(one_phase=True): one call, as before).
    early = early and len(subs) == 1 and not opt.one_phase      # several blocks: the one call enqueues them from a thread each
    for b, (lo, hi, sub) in enumerate(subs):
        c["b"] = b
        _fill_block_opacity(table.blocks[b], sub, lo, hi, c)
    job, keep = drv.make_job(nlayer, plan, factors, linear, 0 if raman == 1 else nlayer, common["stream"],
                             common["delta_eddington"], do_r, do_t, ng, nt, geom["ubar0"], geom["ubar1"], geom["cos_theta"],
                             geom["gweight"], geom["tweight"], toon["single_phase"], toon["multi_phase"],
                             toon["toon_coefficients"], frac_a, frac_b, frac_c, common["TTHG_params"]["constant_back"],
                             common["TTHG_params"]["constant_forward"], 0.0, atm.level["temperature"], atm.level["pressure"],
                             atm.hard_surface, sh=inp["approx"]["rt_params"]["SH"] if is_sh else None, sh_top=sh_top,
                             gauss_wts=opa.gauss_wts if ck else None,
                             after_opacity=(lambda j: drv.enqueue(table, j, phase=1)) if early else None)
    for b, (lo, hi, sub) in enumerate(subs):
        c["b"] = b
        _fill_block_legs(table.blocks[b], sub, lo, hi, c)
    p = _prepared(c, job, keep, key[1:-1])
    p["phase"] = 2 if early else 0
    return p


def _in_scope_3d(inp, opa, legs, opt):
    """The 3-D calls the C driver covers: what ``Spectrum._plan_3d`` sends through ONE fused gas + mixing launch over all
    facets (facet-major planes) -- reflected and / or thermal Toon, monochromatic resident tables, no cloud or cloud tables
    on their own wavenumber grid, Raman off or Pollack -- without the A/B switches that choose another layout."""
    if (opt.no_driver or opt.all_planes or opt.facet_loop or opt.facet_fastest or opt.host_regrid or opt.raman_planes
            or opt.unfused_opacity or opt.regrid_planes):
        return False
    if not legs or not legs <= {"reflected", "thermal"}:
        return False
    if (opa.ngauss != 1 or getattr(opa, "on_fly", False) or inp["test_mode"] is not None or not hasattr(opa, "_cia")
            or not hasattr(opa, "_ray") or inp["approx"].get("get_lvl_flux", False)
            or inp["approx"]["rt_method"] == "SH" or inp["clouds"].get("do_holes", False)):
        return False                # (SH / do_holes in 3-D: Spectrum warns and ignores them, as the reference -- left to it)
    return inp["approx"]["rt_params"]["common"]["raman"] in (1, 2) and inp["atmosphere"]["exclude_mol"] == 1


def prepare_3d(bundle, opa, subs, calculation, opt, slot=None):
    """``prepare`` for ``dimension='3d'`` (reference justdoit.py:407-516): the facet-form set-up and the tall plan once
    for all wavelength blocks, a block table with facet-major planes, a job with ``nfacets``."""
    from .spectrum import setup_facets_3d
    inp = bundle.inputs
    legs = set(calculation.split("+"))
    if not _in_scope_3d(inp, opa, legs, opt):
        return None
    common, toon, geom = inp["approx"]["rt_params"]["common"], inp["approx"]["rt_params"]["toon"], inp["disco"]
    raman = common["raman"]
    wno, nwno = opa.wno, opa.nwno
    ng, nt = geom["num_gangle"], geom["num_tangle"]
    nfac = ng * nt
    cld3 = inp["clouds"].get("profile_3d")
    atm_f, atm, tlev3, plev3 = setup_facets_3d(inp, opa, wno, ng, nt)
    plan, factors = optics.tall_plan(atm_f, opa, nfac, 1)
    if plan.get("premixed"):
        return None
    nlevel, nlayer = atm.c.nlevel, atm.c.nlayer
    tabs_sig = stamp3 = None
    if cld3 is not None:
        if not isinstance(cld3, dict) or cld3.get("wavenumber") is None:
            return None             # cloud arrays on the opacity grid: the facet-fastest mixing launch (Spectrum)
        # ONE digest of the tables per call (0.5 ms for 64 facets x 90 x 196 x 3), whatever the number of blocks
        stamp3 = optics._table_fingerprint([cld3[k] for k in ("opd", "w0", "g0")], cld3["wavenumber"])
        first = optics._facet_major_cloud_tables(cld3, nlayer, nfac, subs[0][2].ctx, stamp=stamp3)
        if first is None:
            return None
        tabs_sig = first[3]
    nmol, ncont, nray = len(plan["molecules"]), len(plan["cia_pairs"]), len(factors[2])
    if (nmol * 56 + ncont * 12 + nray * 8 + 8) * nfac * nlayer > 3600 * 1024:
        return None                 # the per-layer tables of the tall atmosphere must fit ONE table slot (one launch)
    do_r, do_t = "reflected" in legs, "thermal" in legs
    linear = opa.query_method == "linear"
    frac_a, frac_b, frac_c = common["TTHG_params"]["fraction"]
    want, th3 = set(), ("dtau_og", "w0_no_raman", "cosb_og")
    if cld3 is None:                # Spectrum._want_3d(clear3=True)
        if do_r:
            want |= {"dtau", "w0"}
        if do_t:
            th3 = ("dtau", "w0" if (raman == 2 and do_r) else "w0_no_raman", None)
    elif do_r:
        want |= set(resident.REFLECTED_PLANES) - {"tau", "tau_og", "gcos2"}
    if do_t:
        want |= {k for k in th3 if k is not None}

    def table_ids(sub):
        mt = sub._mol_log if linear else sub._mol_raw
        return tuple(id(mt[m]) for m in plan["molecules"]) + tuple(id(sub._cia[p]) for p in plan["cia_pairs"])
    key = ("3d", tuple((lo, hi, id(sub)) + table_ids(sub) for lo, hi, sub in subs), nlayer, ng, nt, tuple(plan["molecules"]),
           tuple(plan["cia_pairs"]), tuple(factors[2]), linear, tuple(sorted(want)), th3, do_r, do_t, slot)
    cache = opa.__dict__.setdefault("_driver_tables", {})
    table = cache.get(key)
    if table is None:
        if len(cache) > (8 if slot is None else 40):
            cache.clear()
        table = cache[key] = drv.BlockTable(subs, nlayer, ng, nt, plan["molecules"], plan["cia_pairs"], factors[2], linear,
                                            want, False, False, do_r, do_t, _constant_planes, facets=nfac, th3=th3)
    c = _call_state(inp, opa, subs, opt, atm, raman, do_r, do_t, table, ng, nt)
    hold = c["hold"]
    for b, (lo, hi, sub) in enumerate(subs):
        c["b"] = b
        k = table.blocks[b]
        _fill_block(k, sub, lo, hi, c)
        if cld3 is not None:        # the tall tables, resident per device (kept on the cloud dictionary by content)
            d_xp, d_tall, _, nin = optics._facet_major_cloud_tables(cld3, nlayer, nfac, sub.ctx, stamp=stamp3)
            hold.append((d_xp, d_tall))
            k.cld_tab_nin, k.cld_tab_xp, k.cld_tab_fp = nin, drv._dev(d_xp), drv._dev(d_tall)
            k.wno = drv._dev(_resident_vector(sub, "wno", sub.wno, hi - lo))
    tl = np.ascontiguousarray(np.asarray(tlev3, dtype=float).reshape(nlevel, nfac).T)
    pv = np.ascontiguousarray(np.asarray(plev3, dtype=float).reshape(nlevel, nfac).T)
    job, keep = drv.make_job(nlayer, plan, factors, linear, 0, common["stream"], common["delta_eddington"], do_r, do_t, ng, nt,
                             geom["ubar0"], geom["ubar1"], geom["cos_theta"], geom["gweight"], geom["tweight"],
                             toon["single_phase"], toon["multi_phase"], toon["toon_coefficients"], frac_a, frac_b, frac_c,
                             common["TTHG_params"]["constant_back"], common["TTHG_params"]["constant_forward"], 0.0, tl, pv,
                             atm.hard_surface, nfacets=nfac)
    return _prepared(c, job, (keep, tabs_sig), key[2:-1])


def finish(p):
    """Second half of ``run``: the results as they arrive (their copies were enqueued with the launches)."""
    table, do_r, do_t, full, nwno, integrals, denom = (p[k] for k in ("table", "do_r", "do_t", "full", "nwno", "integrals", "denom"))
    wno, stellar, inp, atm, opa = (p[k] for k in ("wno", "stellar", "inp", "atm", "opa"))
    returns = {}
    out = {"wavenumber": wno}
    if do_r:
        drv.collect(table, 1)
        returns["albedo"] = full["albedo"][:nwno]
        if integrals:
            returns["bond_integral"] = (full["albedo"][nwno], denom)
        _post_reflected(out, returns, wno, stellar, inp["star"]["semi_major"], atm.planet.radius, opa)
    if do_t:
        drv.collect(table, 2)
        returns["thermal"] = full["thermal"][:nwno]
        if integrals:
            returns["teff_integral"] = full["thermal"][nwno]
        _post_thermal(out, returns, wno, stellar, inp["star"]["radius"], atm.planet.radius, opa)
    return _post_final(out, returns)
