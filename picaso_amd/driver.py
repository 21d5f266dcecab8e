"""ctypes side of ``picaso_toon_spectrum_blocks`` (``csrc/driver.hip``): one C call enqueues every launch of a 1-D
(Toon or SH) or a 3-D spectrum -- gas stage, ``compute_opacity``, reflected and thermal solvers with their fused disk sums -- for
every wavelength block of the spectrum (reference sequence: justdoit.py:236-385; its fan-out over worker
processes: justdoit.py:4774).

``BlockTable`` holds what does not change between the spectra of a retrieval: per block the pointers into its
resident opacity tables and a workspace of planes allocated once.  ``run`` fills the per-call job (per-layer table
rows, weights and coefficients, level temperatures, geometry, options) and makes the call.  The C function only
chains the library's own entry points, so the results are those of the call-by-call path in ``justdoit.picaso``."""
import ctypes

import numpy as np

from . import _lib
from .device import DeviceArray, PinnedArray

_dp = ctypes.POINTER(ctypes.c_double)
_dpp = ctypes.POINTER(_dp)
_ip = ctypes.POINTER(ctypes.c_int)

OUT_NAMES = ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray", "gcos2", "dtau_og", "tau_og", "w0_og", "cosb_og",
             "w0_no_raman", "f_deltaM")                                 # order of picaso_compute_opacity_ck_dev
REFL_NAMES = ("dtau", "tau", "w0", "cosb", "gcos2", "ftau_cld", "ftau_ray", "dtau_og", "tau_og", "w0_og", "cosb_og")
SH_NAMES = ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray", "f_deltaM", "dtau_og", "tau_og", "w0_og", "cosb_og")


class Block(ctypes.Structure):
    _fields_ = [("ctx", ctypes.c_void_p), ("tctx", ctypes.c_void_p), ("nwno", ctypes.c_int), ("col0", ctypes.c_long),
                ("mol_tabs", _dpp), ("cont_tabs", _dpp), ("ray_tabs", _dpp),
                ("cld_opd", _dp), ("cld_w0", _dp), ("cld_g0", _dp),
                ("cld_host_opd", _dp), ("cld_host_w0", _dp), ("cld_host_g0", _dp), ("cld_host_pitch", ctypes.c_long),
                ("cld_work_opd", _dp), ("cld_work_w0", _dp), ("cld_work_g0", _dp),
                ("raman", _dp), ("surf_reflect", _dp), ("F0PI", _dp), ("wno", _dp),
                ("taugas", _dp), ("tauray", _dp), ("planes", _dp * 13), ("refl_planes", _dp * 11),
                ("th_dtau", _dp), ("th_w0", _dp), ("th_cosb", _dp),
                ("xint", _dp), ("albedo", _dp), ("flux", _dp), ("disk", _dp),
                ("albedo_host", _dp), ("thermal_host", _dp), ("trapz_d", _dp), ("trapz_dr", _dp), ("stellar", _dp),
                ("albedo_pin", _dp), ("thermal_pin", _dp), ("albedo_mark", ctypes.c_void_p), ("thermal_mark", ctypes.c_void_p),
                ("cld_tab_nin", ctypes.c_int), ("cld_tab_xp", _dp), ("cld_tab_fp", _dp)]


class Job(ctypes.Structure):
    _fields_ = [("nlayer", ctypes.c_int), ("mol_mode", ctypes.c_int), ("nmol", ctypes.c_int),
                ("cont_interp", ctypes.c_int), ("ncont", ctypes.c_int), ("nray", ctypes.c_int),
                ("mol_rows", _ip), ("mol_wts", _dp), ("mol_fac", _dp), ("cont_rows", _ip), ("cont_wts", _dp),
                ("cont_fac", _dp), ("ray_fac", _dp), ("raman_rows", ctypes.c_int), ("raman_const", ctypes.c_double),
                ("test_mode", ctypes.c_int), ("delta_eddington", ctypes.c_int), ("stream", ctypes.c_int),
                ("do_reflected", ctypes.c_int), ("do_thermal", ctypes.c_int), ("numg", ctypes.c_int),
                ("numt", ctypes.c_int), ("ubar0", _dp), ("ubar1", _dp), ("cos_theta", ctypes.c_double),
                ("gweight", _dp), ("tweight", _dp), ("single_phase", ctypes.c_int), ("multi_phase", ctypes.c_int),
                ("toon_coefficients", ctypes.c_int), ("frac_a", ctypes.c_double), ("frac_b", ctypes.c_double),
                ("frac_c", ctypes.c_double), ("constant_back", ctypes.c_double), ("constant_forward", ctypes.c_double),
                ("b_top", ctypes.c_double), ("tlevel", _dp), ("plevel", _dp), ("hard_surface", ctypes.c_int),
                ("rt_method", ctypes.c_int), ("sh_w_single_form", ctypes.c_int), ("sh_w_multi_form", ctypes.c_int),
                ("sh_psingle_form", ctypes.c_int), ("sh_w_single_rayleigh", ctypes.c_int),
                ("sh_w_multi_rayleigh", ctypes.c_int), ("sh_psingle_rayleigh", ctypes.c_int),
                ("sh_single_form", ctypes.c_int), ("sh_cloud_free_above", ctypes.c_int), ("nfacets", ctypes.c_int),
                ("ngauss", ctypes.c_int), ("gauss_wts", _dp)]


def _dev(x):
    """DeviceArray / raw address / None -> POINTER(c_double)"""
    if x is None:
        return None
    return ctypes.cast(ctypes.c_void_p(int(x.addr if isinstance(x, DeviceArray) else x)), _dp)


def _host(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _table_ptrs(devs):
    arr = (_dp * max(1, len(devs)))()
    for i, d in enumerate(devs):
        arr[i] = _dev(d)
    return arr


class BlockTable:
    """The blocks of one opacity object (``subs``: [(lo, hi, shard opacity object)]; one entry covering the whole grid
    for a single-GPU spectrum) for one signature of the call -- molecule / continuum / Rayleigh species in table order,
    layer count, legs, which planes are written.  Workspaces are allocated once and reused by every later spectrum
    of the same signature: every access is ordered on the blocks' streams."""

    def __init__(self, subs, nlayer, ng, nt, mol_names, cia_pairs, ray_names, linear, want, lean, host_cloud,
                 do_reflected, do_thermal, const_planes, derive=False, sh=False, facets=0, th3=None, ngauss=1):
        self.subs, self.n = subs, len(subs)
        self.blocks = (Block * self.n)()
        self.keep = []                                   # DeviceArrays and pointer tables the structs point into
        self.want = tuple(want)
        for b, (lo, hi, sub) in enumerate(subs):
            k = self.blocks[b]
            ctx, nw = sub.ctx, hi - lo
            k.ctx = ctx.value if hasattr(ctx, "value") else ctx
            k.tctx = None
            k.nwno, k.col0 = nw, lo
            # premixed k-tables (ngauss > 1): ONE table of nwno * ngauss columns; planes (rows, nwno, ngauss), Gauss index fastest
            tabs = [sub._kappa] if ngauss > 1 else [(sub._mol_log if linear else sub._mol_raw)[m] for m in mol_names]
            ncolg = nw * ngauss
            ctabs, rtabs = [sub._cia[p] for p in cia_pairs], [sub._ray[m] for m in ray_names]
            mt, ct, rt = _table_ptrs(tabs), _table_ptrs(ctabs), _table_ptrs(rtabs)
            self.keep += [mt, ct, rt, tabs, ctabs, rtabs]    # the tables themselves too: the structs hold raw addresses
            k.mol_tabs, k.cont_tabs, k.ray_tabs = ctypes.cast(mt, _dpp), ctypes.cast(ct, _dpp), ctypes.cast(rt, _dpp)
            if facets:
                # a 3-D block (Job.nfacets): facet-major planes (nfacets, nlayer, nw) from the fused launch over the tall
                # atmosphere; no TAUGAS / TAURAY workspace (that launch keeps the sums in registers), no level planes
                self._facet_block(k, ctx, nw, nlayer, ng, nt, facets, want, do_reflected, do_thermal, th3)
                continue
            tg, tr = DeviceArray((nlayer, ncolg), ctx), DeviceArray((nlayer, nw), ctx)
            self.keep += [tg, tr]
            k.taugas, k.tauray = _dev(tg), _dev(tr)
            pl = {}
            for i, name in enumerate(OUT_NAMES):
                if name in want:
                    pl[name] = DeviceArray((nlayer + 1 if name in ("tau", "tau_og") else nlayer, ncolg), ctx)
                    self.keep.append(pl[name])
                    k.planes[i] = _dev(pl[name])
            rpl = pl
            if sh and lean:                              # SH, no cloud: the launch reads dtau and w0 (thermal: + cosb_og = 0)
                zero, one, half = const_planes(sub, nlayer, nw)
                rpl = {"dtau": pl["dtau"], "w0": pl["w0"]}
                pl = dict(pl, cosb_og=zero)
            elif sh:
                pass                                     # the planes as written (level planes may be left out: derived)
            elif lean and derive:                        # the reflected kernel re-derives all but dtau and w0
                zero, one, half = const_planes(sub, nlayer, nw)
                rpl = {"dtau": pl["dtau"], "w0": pl.get("w0")}
                pl.update(dtau_og=pl["dtau"], cosb_og=zero)
                if "w0_no_raman" not in pl and "w0" in pl:
                    pl["w0_no_raman"] = pl["w0"]
            elif lean:                                   # justdoit.picaso: aliases and constants of a cloud-free atmosphere
                zero, one, half = const_planes(sub, nlayer, nw)
                pl.update(dtau_og=pl["dtau"], cosb=zero, cosb_og=zero, ftau_cld=zero, ftau_ray=one, gcos2=half)
                if "tau" in pl:
                    pl.update(tau_og=pl["tau"], w0_og=pl["w0"])
                if "w0_no_raman" not in pl and "w0" in pl:
                    pl["w0_no_raman"] = pl["w0"]
            if do_reflected:
                for i, name in enumerate(SH_NAMES if sh else REFL_NAMES):
                    k.refl_planes[i] = _dev(rpl.get(name))       # None: left out, re-derived in the kernel
                x, a = DeviceArray((ng, nt, nw), ctx), DeviceArray((nw + 1,), ctx)     # [nw]: the Bond-albedo integral
                pin = PinnedArray((nw + 1,), ctx)           # the result copy is enqueued with the launches
                self.keep += [x, a, pin]
                k.xint, k.albedo, k.albedo_pin = _dev(x), _dev(a), ctypes.cast(ctypes.c_void_p(pin.addr), _dp)
            if do_thermal and sh:                        # get_thermal_SH reads dtau, w0 and cosb_og (spectrum._thermal_sh)
                k.th_dtau, k.th_w0, k.th_cosb = _dev(pl["dtau"]), _dev(pl["w0"]), _dev(pl["cosb_og"])
            elif do_thermal:
                k.th_dtau, k.th_w0, k.th_cosb = _dev(pl["dtau_og"]), _dev(pl["w0_no_raman"]), _dev(pl["cosb_og"])
            if host_cloud:
                cw = [DeviceArray((nlayer, nw), ctx) for _ in range(3)]
                self.keep += cw
                k.cld_work_opd, k.cld_work_w0, k.cld_work_g0 = (_dev(c) for c in cw)
        self.thermal_ws = {}                              # thermal outputs live on the thermal leg's context

    def _facet_block(self, k, ctx, nw, nlayer, ng, nt, facets, want, do_reflected, do_thermal, th3):
        if {"tau", "tau_og"} & set(want):
            raise ValueError("3-D blocks: the level planes are running sums inside the solvers")
        pl = {}
        for i, name in enumerate(OUT_NAMES):
            if name in want:
                pl[name] = DeviceArray((facets, nlayer, nw), ctx)
                self.keep.append(pl[name])
                k.planes[i] = _dev(pl[name])
        if do_reflected:
            for i, name in enumerate(REFL_NAMES):
                k.refl_planes[i] = _dev(pl.get(name))
            x, a = DeviceArray((ng, nt, nw), ctx), DeviceArray((nw + 1,), ctx)
            pin = PinnedArray((nw + 1,), ctx)
            self.keep += [x, a, pin]
            k.xint, k.albedo, k.albedo_pin = _dev(x), _dev(a), ctypes.cast(ctypes.c_void_p(pin.addr), _dp)
        if do_thermal:
            k.th_dtau, k.th_w0 = _dev(pl[th3[0]]), _dev(pl[th3[1]])
            k.th_cosb = _dev(pl[th3[2]]) if th3[2] else None

    def thermal_workspace(self, b, tctx, ng, nt):
        """flux / disk of block b on the context the thermal leg runs on (allocated once per context)."""
        key = (b, getattr(tctx, "value", tctx))
        ws = self.thermal_ws.get(key)
        if ws is None:
            nw = self.blocks[b].nwno
            ws = self.thermal_ws[key] = (DeviceArray((ng, nt, nw), tctx), DeviceArray((nw + 1,), tctx),    # [nw]: T_eff integral
                                         PinnedArray((nw + 1,), tctx))
        return ws


def make_job(nlayer, plan, factors, linear, raman_rows, stream, delta_eddington, do_reflected, do_thermal, ng, nt, ubar0,
             ubar1, cos_theta, gweight, tweight, single_phase, multi_phase, toon_coefficients, frac_a, frac_b, frac_c,
             constant_back, constant_forward, b_top, tlevel, plevel, hard_surface, sh=None, sh_top=0, nfacets=0,
             gauss_wts=None, after_opacity=None):
    """The per-call half: (Job, the numpy arrays it points into).  ``plan`` = ``opa._plan`` (table rows and weights per
    molecule and layer, CIA rows), ``factors`` = ``optics._layer_factors`` (per-layer coefficients of the sums).
    ``after_opacity``: called with the Job as soon as the fields the opacity stage reads are set (``enqueue(..., phase=1)``
    from there puts the gas kernel on the stream while the other half of the job is still being filled)."""
    mol_fac, cont_fac, ray_names, ray_fac = factors
    nmol, ncont = len(plan["molecules"]), len(plan["cia_pairs"])
    premixed = bool(plan.get("premixed"))         # k-tables: cia_rows / cia_wts (nlayer, 2), the bracketing temperatures
    # ---- the opacity stage's half ----
    keep = dict(
        rows=np.ascontiguousarray(plan["rows"], dtype=np.int32), wts=_lib.f64(plan["wts"]), mol_fac=_lib.f64(mol_fac),
        cont_rows=np.ascontiguousarray(np.repeat(plan["cia_rows"][None], max(ncont, 1), axis=0), dtype=np.int32),
        cont_wts=_lib.f64(np.repeat(plan["cia_wts"][None], max(ncont, 1), axis=0)) if premixed else None,
        cont_fac=_lib.f64(cont_fac), ray_fac=_lib.f64(ray_fac))
    j = Job()
    j.nlayer, j.mol_mode, j.nmol, j.cont_interp, j.ncont, j.nray = (nlayer, (2 if premixed else (1 if linear else 0)), nmol,
                                                                    (1 if premixed else 0), ncont, len(ray_names))
    j.mol_rows = keep["rows"].ctypes.data_as(_ip) if nmol else None
    j.mol_wts, j.mol_fac = (_host(keep["wts"]), _host(keep["mol_fac"])) if nmol else (None, None)
    j.cont_rows = keep["cont_rows"].ctypes.data_as(_ip) if ncont else None
    j.cont_wts, j.cont_fac = (_host(keep["cont_wts"]) if (premixed and ncont) else None), (_host(keep["cont_fac"]) if ncont else None)
    j.ray_fac = _host(keep["ray_fac"]) if len(ray_names) else None
    j.raman_rows, j.raman_const = raman_rows, 0.99999
    j.test_mode, j.delta_eddington, j.stream = 0, (1 if delta_eddington else 0), stream
    j.do_reflected, j.do_thermal, j.numg, j.numt = int(do_reflected), int(do_thermal), ng, nt
    j.rt_method, j.nfacets = 0, int(nfacets)    # nfacets > 0: plan / factors of the tall atmosphere, tlevel / plevel (nfacets, nlevel)
    j.ngauss, j.gauss_wts = 1, None
    if gauss_wts is not None and len(gauss_wts) > 1:      # premixed k-tables: the Gauss-point loop inside the solver calls
        keep["gauss_wts"] = _lib.f64(gauss_wts)
        j.ngauss, j.gauss_wts = int(len(gauss_wts)), _host(keep["gauss_wts"])
    if after_opacity is not None:
        after_opacity(j)
    # ---- the legs' half ----
    keep.update(u0=_lib.f64(ubar0, (ng, nt)), u1=_lib.f64(ubar1, (ng, nt)), gw=_lib.f64(gweight), tw=_lib.f64(tweight),
                tl=_lib.f64(tlevel), pl=_lib.f64(plevel))
    j.ubar0, j.ubar1, j.cos_theta = _host(keep["u0"]), _host(keep["u1"]), float(cos_theta)
    j.gweight, j.tweight = _host(keep["gw"]), _host(keep["tw"])
    j.single_phase, j.multi_phase, j.toon_coefficients = int(single_phase), int(multi_phase), int(toon_coefficients)
    j.frac_a, j.frac_b, j.frac_c = float(frac_a), float(frac_b), float(frac_c)
    j.constant_back, j.constant_forward, j.b_top = float(constant_back), float(constant_forward), float(b_top)
    j.tlevel, j.plevel, j.hard_surface = _host(keep["tl"]), _host(keep["pl"]), int(hard_surface)
    if sh is not None:                     # inputs["approx"]["rt_params"]["SH"]: the spherical-harmonics solvers
        j.rt_method = 1
        j.sh_w_single_form, j.sh_w_multi_form, j.sh_psingle_form = (int(sh[k]) for k in ("w_single_form", "w_multi_form", "psingle_form"))
        j.sh_w_single_rayleigh, j.sh_w_multi_rayleigh, j.sh_psingle_rayleigh = (
            int(sh[k]) for k in ("w_single_rayleigh", "w_multi_rayleigh", "psingle_rayleigh"))
        j.sh_single_form, j.sh_cloud_free_above = int(sh["single_form"]), int(sh_top)
    return j, keep


def enqueue(table, job, phase=0):
    """``phase`` 0: the whole spectrum in one call; 1: the opacity stage alone, 2: everything behind it
    (``picaso_toon_spectrum_phase``, 1-D blocks: what ``make_job(after_opacity=...)`` is for)."""
    if phase:
        rc = _lib.load().picaso_toon_spectrum_phase(ctypes.c_int(table.n), table.blocks, ctypes.byref(job), ctypes.c_int(phase))
    else:
        rc = _lib.load().picaso_toon_spectrum_blocks(ctypes.c_int(table.n), table.blocks, ctypes.byref(job))
    _lib.check(rc, table.subs[0][2].ctx)


def collect(table, which):
    _lib.check(_lib.load().picaso_toon_spectrum_collect(ctypes.c_int(table.n), table.blocks, ctypes.c_int(which)),
               table.subs[0][2].ctx)


def abandon(table):
    """Drop result copies nobody will collect (an exception between ``enqueue`` and ``collect``)."""
    _lib.load().picaso_toon_spectrum_abandon(ctypes.c_int(table.n), table.blocks)

