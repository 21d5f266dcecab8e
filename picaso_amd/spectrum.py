"""One spectrum on one context: plan -> enqueue -> finish (the body of the reference's ``picaso()``, justdoit.py:65-621).

``justdoit.picaso`` is the entry point and decides between the one-C-call driver (csrc/driver.hip), the multi-GPU split
(``devices=``) and this module's ``Spectrum``; ``spectrum_batch`` and ``phase_curve`` enqueue several ``Spectrum`` objects
before they finish the first.  The helpers here are the resident per-wavelength vectors, the ATMSETUP sequence, the
launch wrappers of the solvers and the spectrum-wide post-processing (Bond albedo, flux ratios, effective temperature).
"""
import copy
import warnings

import numpy as np

from . import _lib, device, fastsetup, optics, resident
from . import options as _options
from .atmsetup import ATMSETUP, CloudTables
from .device import DeviceArray


def _interp_axis(x_new, x_old, arr, axis, period=None):
    """Linear interpolation of ``arr`` along ``axis`` from the grid ``x_old`` to ``x_new`` (end values held outside the
    grid).  ``x_old`` is sorted here if need be; repeated abscissae (a longitude axis holding both -180 and 180) keep
    their first entry.  ``period``: the axis is periodic (longitude, 360): points beyond either end are interpolated
    across the seam instead of being clamped to the end value."""
    x_old = np.asarray(x_old, dtype=float)
    arr = np.asarray(arr)
    if np.any(np.diff(x_old) <= 0):
        x_old, first = np.unique(x_old, return_index=True)          # sorted, duplicates dropped
        arr = np.take(arr, first, axis=axis)
    if period is not None and x_old.size > 1:
        if x_old[-1] - x_old[0] >= period:                          # both ends of the seam present: one of them is the other
            keep = x_old < x_old[0] + period
            x_old, arr = x_old[keep], np.compress(keep, arr, axis=axis)
        x_old = np.concatenate([[x_old[-1] - period], x_old, [x_old[0] + period]])
        arr = np.concatenate([np.take(arr, [-1], axis=axis), arr, np.take(arr, [0], axis=axis)], axis=axis)
        x_new = x_old[1] + np.mod(np.asarray(x_new, dtype=float) - x_old[1], period)
    if x_old.size == 1:
        return np.repeat(arr, len(x_new), axis=axis)
    x = np.clip(np.asarray(x_new, dtype=float), x_old[0], x_old[-1])
    j = np.clip(np.searchsorted(x_old, x, side="right") - 1, 0, x_old.size - 2)
    t = (x - x_old[j]) / (x_old[j + 1] - x_old[j])
    shape = [1] * arr.ndim
    shape[axis] = -1
    t = t.reshape(shape)
    return np.take(arr, j, axis=axis) * (1.0 - t) + np.take(arr, j + 1, axis=axis) * t


def _resident_vector(opa, name, value, nwno):
    """Per-wavelength vector (or a scalar broadcast to one) in HBM, kept on the opacity object while its
    content does not change: a retrieval calls spectrum() with the same grid, stellar spectrum and
    surface reflectivity thousands of times (3 x 0.8 MB of H2D per call at 1e5 wavelengths)."""
    cache = opa.__dict__.setdefault("_resident_vectors", {})
    hit = cache.get(name)
    if np.ndim(value) == 0:                        # a scalar: compared as one (no 1e5-element array per call)
        key = float(value)
        # (the tag of an array entry is the ndarray itself: compare tuples only)
        if hit is not None and isinstance(hit[2], tuple) and hit[2] == ("scalar", key, nwno):
            return hit[1]
        a = np.full(nwno, key)
        tag = ("scalar", key, nwno)
    else:
        if name == "wno" and hit is not None and not isinstance(hit[2], tuple) and hit[2] is value:    # the opacity object's own grid: never edited
            return hit[1]
        a = np.ascontiguousarray(np.zeros(nwno) + np.asarray(value, dtype=float))
        if hit is not None and hit[0] is not None and hit[0].shape == a.shape and np.array_equal(hit[0], a):
            cache[name] = (hit[0], hit[1], value)
            return hit[1]
        tag = value
    d = DeviceArray.from_host(a, opa.ctx)
    cache[name] = (a.copy(), d, tag)
    return d


def _ones(opa, nwno):
    """``np.zeros(nwno) + 1.0`` (the reference's F0PI without a star, justdoit.py:174-175), kept on the opacity object
    and read-only: nothing on the path writes to F0PI."""
    hit = opa.__dict__.get("_ones")
    if hit is None or hit.shape != (nwno,):
        hit = np.zeros(nwno) + 1.0
        hit.flags.writeable = False
        opa.__dict__["_ones"] = hit
    return hit


def _cloud_free_top(inp, nlayer):
    """Number of layers above the cloud deck: the first layer whose cloud profile rows hold any optical depth or any
    asymmetry (COSB is the cloud's g0 itself, optics.py:338, so a g0 without optical depth still delta-scales the layer).
    Read off the profile AS GIVEN (linear regridding keeps a zero row zero); tables larger than 2e5 numbers are not
    scanned (0: no statement) -- the scan would cost more than it saves."""
    prof = inp["clouds"]["profile"]
    if prof is None:
        return nlayer
    busy = np.zeros(nlayer, dtype=bool)
    for k in ("opd", "g0"):
        v = np.asarray(prof[k], dtype=np.float64)
        if v.ndim == 0:
            return 0 if v != 0 else nlayer
        if v.size > 200000 or v.size % nlayer:
            return 0
        busy |= (v.reshape(nlayer, -1) != 0).any(axis=1)
    return int(np.argmax(busy)) if busy.any() else nlayer


def _constant_planes(opa, nlayer, nwno):
    """Resident ``(nlayer, nwno)`` planes of 0, 1 and 0.5, kept on the opacity object: what ``compute_opacity``
    writes into cosb / cosb_og / ftau_cld, ftau_ray and gcos2 for an atmosphere without cloud."""
    cache = opa.__dict__.setdefault("_const_planes", {})
    key = (nlayer, nwno)
    if key not in cache:
        cache[key] = (DeviceArray.zeros((nlayer, nwno), opa.ctx),
                      DeviceArray.from_host(np.ones((nlayer, nwno)), opa.ctx),
                      DeviceArray.from_host(np.full((nlayer, nwno), 0.5), opa.ctx))
    return cache[key]


def _setup_atmosphere(inp, opa, wno, profile=None, cloud_profile=None):
    """ATMSETUP sequence of the reference's ``picaso()`` (justdoit.py:180-243) for the 1-D profile
    or, in the 3-D path, for one facet's profile (``atm_1d.disect(g,t)``, justdoit.py:446-449)."""
    if profile is None:                    # the whole set-up in one C call where it applies (fastsetup.py)
        fast = fastsetup.setup(inp, opa, wno)
        if fast is not None:
            return fast
    elif cloud_profile is None and all(getattr(v, "ndim", 0) == 2 for v in profile.values()):      # facet form (3-D path)
        fast = fastsetup.setup_facets(inp, opa, wno, profile)
        if fast is not None:
            return fast
    cfg = inp
    if profile is not None:
        cfg = dict(inp)
        cfg["atmosphere"] = dict(inp["atmosphere"], profile=profile)
        cfg["clouds"] = dict(inp["clouds"], profile=cloud_profile)
    atm = ATMSETUP(cfg)
    atm.surf_reflect = inp.get("surface_reflect", 0)
    atm.hard_surface = inp.get("hard_surface", 0)
    atm.wavenumber = wno
    atm.planet.gravity = inp["planet"]["gravity"]
    atm.planet.radius = inp["planet"]["radius"]
    atm.planet.mass = inp["planet"]["mass"]
    atm.get_lvl_flux = inp["approx"].get("get_lvl_flux", False)
    atm.get_profile()
    atm.get_mmw()
    atm.get_density()
    atm.get_altitude(p_reference=inp["approx"]["p_reference"])
    atm.get_column_density()
    atm.get_needed_continuum(opa.rayleigh_molecules, opa.avail_continuum)
    atm.get_clouds(wno)
    no_opa = [m for m in atm.molecules if m not in opa.molecules]
    if no_opa:
        atm.add_warnings("I found chemistry for these but I do not have computed individual line "
                         "opacities (not including continuum) for: " + ",".join(no_opa))
    atm.molecules = np.array([m for m in atm.molecules if m not in no_opa])
    return atm


def _atmosphere_block(atm0, lo, hi, wno):
    """One wavelength block's view of an ATMSETUP that was set up once for the whole grid: everything but the
    cloud tables and the wavenumbers is per layer / level and shared; the cloud arrays are column slices (views)."""
    atm = copy.copy(atm0)
    atm.wavenumber = wno
    atm.layer = dict(atm0.layer)
    cld = atm0.layer["cloud"]
    if isinstance(cld, CloudTables):          # tables on their own grid: the block regrids its own columns
        atm.layer["cloud"] = cld.columns(lo, hi)
        atm.layer["cloud"].wno = wno
    else:
        atm.layer["cloud"] = {k: v[:, lo:hi] for k, v in cld.items()}
    sr = atm0.surf_reflect
    if np.ndim(sr) > 0 and np.size(sr) == np.size(atm0.wavenumber):
        atm.surf_reflect = np.ascontiguousarray(np.asarray(sr, dtype=float)[lo:hi])
    return atm


def _reflected_3d_fm(ctx, nlevel, nwno, ng, nt, planes, rs, ubar0, ubar1, cos_theta, F0PI, single_phase, multi_phase,
                     frac_a, frac_b, frac_c, constant_back, constant_forward, xint, gweight, tweight, albedo):
    """``resident.reflected_3d`` for facet-major planes (one spectrum through ``reflected_3d_fm_batch``)."""
    resident.reflected_3d_fm_batch(ctx, nlevel, nwno, ng, nt, [planes], [rs], np.asarray(ubar0, dtype=float).reshape(1, ng, nt),
                                   np.asarray(ubar1, dtype=float).reshape(1, ng, nt), np.array([cos_theta], dtype=float),
                                   [F0PI], single_phase, multi_phase, frac_a, frac_b, frac_c, constant_back,
                                   constant_forward, [xint], gweight, tweight, [albedo])


def _fetch(prefetched, key, dev, returns=None, integral=None):
    """Host copy of the resident result ``dev``: the pinned block ``finish.prefetch`` put on the stream when there is
    one (wait for that copy only), a synchronous copy otherwise.  A prefetched block one longer than the result carries
    the result's spectrum-wide integral in its last element: stored as ``returns[integral]``."""
    hit = prefetched.pop(key, None)
    if hit is None:
        return dev.to_host()
    p, denom = hit
    a = p.wait()
    out = a[:dev.size].copy()
    if a.size > dev.size:
        returns[integral] = a[dev.size] if denom is None else (a[dev.size], denom)
    p.free()
    return out




def setup_facets_3d(inp, opa, wno, ng, nt):
    """All facets in ONE facet-form ATMSETUP ((nlevel, nfacets) columns; the reference builds one per facet,
    justdoit.py:437-449), the atmosphere of facet (0, 0) (sizes, surface, full_output) and the level tables the thermal
    solver takes: ``(atm_f, atm, tlev3, plev3)``."""
    prof3 = inp["atmosphere"]["profile_3d"]
    nfac, nlv = ng * nt, len(prof3["pressure"])
    prof_f = {}
    for k, v in prof3.items():
        if k == "temperature":
            prof_f[k] = np.ascontiguousarray(np.broadcast_to(v.reshape(nlv, -1), (nlv, nfac)))
        else:
            prof_f[k] = v.reshape(nlv, -1)            # (nlevel, 1) shared or (nlevel, nfacets)
    atm_f = _setup_atmosphere(inp, opa, wno, prof_f, None)
    atm = _setup_atmosphere(inp, opa, wno, {k: (v if v.ndim == 1 else v[:, 0, 0]) for k, v in prof3.items()}, None)
    tlev3 = atm_f.level["temperature"].reshape(nlv, ng, nt)
    plev3 = np.ascontiguousarray(np.broadcast_to(atm_f.level["pressure"].reshape(nlv, 1, 1), (nlv, ng, nt)))
    return atm_f, atm, tlev3, plev3


# ------------------------------------------------------------------------------------------------
# one spectrum: plan -> enqueue -> finish
# ------------------------------------------------------------------------------------------------
class Spectrum:
    """One ``picaso()`` call (reference justdoit.py:65-621) as three stages shared by every combination of solver
    (Toon / spherical harmonics), dimension (1-D / 3-D), opacity kind (monochromatic / correlated-k), patchy clouds, level
    fluxes, wavelength blocks (``devices=``) and batched launches (``spectrum_batch`` / ``phase_curve``):

    ``plan()``     host only -- ATMSETUP, table rows and weights, WHICH opacity planes the legs will read
                   (reference :180-243, :437-449) -- then the opacity launches that write them (:236-252, :444-471);
    ``enqueue()``  every leg's solver launches, in the reference's order reflected -> thermal -> transmission
                   (:254-405, :488-516).  The correlated-k loop is the column axis of ONE launch per leg
                   (``resident.*_ck``; csrc/ckloop.hip) for every solver, the patchy-cloud blend a second launch + axpby
                   for the solver the reference blends (Toon 1-D);
    ``__call__()`` results back, spectrum-wide integrals, the output dictionary (:527-599).  With ``defer=True`` the caller
                   gets the object itself and calls it later (``prefetch`` puts the result copies on the stream now).

    A combination is a composition of these pieces, not a branch of its own: SH x correlated-k and 3-D x correlated-k
    (round 4: ``raise``) are ``_solver`` picking the ``*_ck`` launcher for planes that carry a Gauss axis."""

    def __init__(self, bundle, opacityclass, dimension="1d", calculation="reflected", full_output=False, as_dict=True,
                 raw=False, shared=None, batch=None, options=None):
        self.opt = _options.current(options)
        self.inp = inp = bundle.inputs
        self.opa = opa = opacityclass
        self.ctx = self.tctx = opa.ctx
        self.dimension, self.calculation = dimension, calculation
        self.full_output, self.as_dict, self.raw, self.shared, self.batch = full_output, as_dict, raw, shared, batch
        self.wno, self.nwno, self.ngauss = opa.wno, opa.nwno, opa.ngauss
        self.gauss_wts = np.asarray(opa.gauss_wts, dtype=float)
        self.common = common = inp["approx"]["rt_params"]["common"]
        self.toon = toon = inp["approx"]["rt_params"]["toon"]
        self.sh = inp["approx"]["rt_params"]["SH"]
        self.frac = tuple(common["TTHG_params"]["fraction"])
        self.tthg = (toon["single_phase"], toon["multi_phase"], *self.frac, common["TTHG_params"]["constant_back"],
                     common["TTHG_params"]["constant_forward"])
        geom = inp["disco"]
        self.ng, self.nt = geom["num_gangle"], geom["num_tangle"]
        self.gweight, self.tweight = geom["gweight"], geom["tweight"]
        self.cos_theta, self.ubar0, self.ubar1 = geom["cos_theta"], geom["ubar0"], geom["ubar1"]
        self.nostar = inp["star"]["database"] == "nostar"
        self.F0PI = (np.zeros(self.nwno) + 1.0) if self.nostar else inp["star"]["relative_flux"]       # justdoit.py:174-175
        stellar = getattr(opa, "unshifted_stellar_spec", None)
        self.stellar = self.F0PI if stellar is None else stellar
        self.b_top = 0.0
        self.sa, self.radius_star = inp["star"]["semi_major"], inp["star"]["radius"]
        self.do_holes = bool(inp["clouds"].get("do_holes", False))
        # the reference's 3-D branch has neither an SH solver, nor patchy clouds, nor level fluxes (justdoit.py:407-516
        # calls get_reflected_3d / get_thermal_3d whatever rt_method says): those settings do nothing there
        self.is_sh = inp["approx"]["rt_method"] == "SH" and dimension == "1d"
        self.lvl_flux = bool(inp["approx"].get("get_lvl_flux", False)) and dimension == "1d" and not self.is_sh
        if self.do_holes and (self.is_sh or dimension == "3d"):
            # justdoit.py:248-252 computes the thinned-cloud planes, but only the Toon 1-D calls read them (:287-305,
            # :346-361); f_deltaM = COSB**stream is the cloud's own g0 and does not change with the thinning
            warnings.warn("do_holes has no effect with rt_method='SH' or dimension='3d' (as in the reference, whose "
                          "patchy-cloud blend exists for the 1-D Toon solver only)", UserWarning)
            self.do_holes = False
        if dimension == "3d" and inp["approx"]["rt_method"] == "SH":
            warnings.warn("dimension='3d' runs the Toon solver (the reference's 3-D branch calls get_reflected_3d / "
                          "get_thermal_3d whatever rt_method says)", UserWarning)
        self.fhole = float(inp["clouds"]["fhole"]) if self.do_holes else None
        self.planes = self.planes_clear = self.rplanes = self.planes3d = None
        self.tlev3 = self.plev3 = None
        self.th3 = ("dtau_og", "w0_no_raman", "cosb_og")
        self.sh_top = 0
        self.xint = None
        self.returns = {"wavenumber": self.wno}
        self.dev = {}             # per-wavelength results still in HBM (the multi-GPU form gathers them with RCCL)
        self.prefetched = {}      # result copies already on the stream (prefetch)
        self.collect = []
        self.keep_alive = []

    # ---------------------------------------------------------------- plan
    def plan(self):
        """Host set-up and the opacity launches: after this the planes every leg reads are enqueued."""
        with _options.use(self.opt):
            if self.dimension == "3d":
                self._plan_3d()
            else:
                self._plan_1d()
        opa, atm = self.opa, self.atm
        self.nlevel, self.nlayer = atm.c.nlevel, atm.c.nlayer
        self.rs = _resident_vector(opa, "surf_reflect", atm.surf_reflect, self.nwno)
        self.d_f0 = _resident_vector(opa, "F0PI", 1.0 if self.nostar else self.F0PI, self.nwno)
        # Spectra (Toon, SH, 3-D) with both legs: the thermal kernels go to a second stream that waits for the
        # opacity planes only, so they run next to the reflected-light kernel (each of the two alone
        # leaves SIMDs idle through its tail) instead of behind it
        if ("reflected" in self.calculation and "thermal" in self.calculation
                and (self.dimension == "1d" or self.batch is None) and self.opt.overlap_legs):
            self.tctx = _lib.aux_context(_lib.device_of(self.ctx))     # one per process and device, shared by every caller
            _lib.ctx_wait(self.tctx, self.ctx)                         # (in a batch: every member's, so the last one covers the launch)
        return self

    def _want_3d(self, clear3):
        """Which planes the 3-D legs read.  Only planes that cannot be re-derived exactly inside the solvers are written
        (each is nfacets x 9 MB at 12 500 wavelengths x 90 layers): the level optical depths are running sums and gcos2
        is 0.5 ftau_ray, so the reflected kernel takes 8 planes instead of 11; without cloud (and outside the test modes)
        cosb = cosb_og = ftau_cld = 0, ftau_ray = 1 and the delta-scaling is the identity, which leaves dtau and w0 --
        and w0_no_raman equals w0 when the Raman factor is the constant 0.99999 (raman='none').  ``all_planes`` writes
        and reads the full set (A/B, tests)."""
        want3 = set()
        calc = self.calculation
        if "reflected" in calc:
            if clear3:
                want3 |= {"dtau", "w0"}
            elif not self.opt.all_planes:
                want3 |= set(resident.REFLECTED_PLANES) - {"tau", "tau_og", "gcos2"}
            else:
                want3 |= set(resident.REFLECTED_PLANES)
        if "thermal" in calc:
            if clear3:
                self.th3 = ("dtau", "w0" if (self.common["raman"] == 2 and "reflected" in calc) else "w0_no_raman", None)
            want3 |= {k for k in self.th3 if k is not None}
        return want3

    def _plan_3d(self):
        """justdoit.py:407-471: one atmosphere per facet and the planes of all facets."""
        inp, opa, wno, opt = self.inp, self.opa, self.wno, self.opt
        ng, nt, ctx, common = self.ng, self.nt, self.ctx, self.common
        prof3 = inp["atmosphere"]["profile_3d"]
        cld3 = inp["clouds"].get("profile_3d")
        if isinstance(cld3, dict) and cld3.get("wavenumber") is not None and not (
                len(cld3["wavenumber"]) == len(wno) and np.array_equal(cld3["wavenumber"], wno)) and opt.host_regrid:
            # a cloud dataset on its own wavenumber grid (clouds_3d(ds)): onto the opacity grid, linear in wavenumber
            # like the reference's per-facet get_clouds -> wavelength.regrid (atmsetup.py:609-622).  Normally on the
            # device (numpy.interp's bits); here the host form of the same interpolation
            cld3 = dict(cld3, **{k: _interp_axis(wno, cld3["wavenumber"], np.asarray(cld3[k], dtype=float), 1)
                                 for k in ("opd", "w0", "g0")})
            cld3.pop("wavenumber")
        clear3 = cld3 is None and inp["test_mode"] is None and not opt.all_planes
        want3 = self._want_3d(clear3)
        co3 = dict(stream=common["stream"], delta_eddington=common["delta_eddington"], test_mode=inp["test_mode"],
                   raman=common["raman"], clouds_3d=cld3, exclude_mol=inp["atmosphere"]["exclude_mol"], want=want3)
        if opt.facet_loop and self.ngauss == 1:               # A/B: one ATMSETUP + one gas launch per facet
            atms = [[_setup_atmosphere(inp, opa, wno, {k: (v if v.ndim == 1 else v[:, g, t]) for k, v in prof3.items()}, None)
                     for t in range(nt)] for g in range(ng)]
            self.atm = atms[0][0]
            self.planes3d = optics.compute_opacity_facets(atms, opa, ng, nt, **co3)
            self.tlev3 = np.stack([np.stack([a_.level["temperature"] for a_ in row], axis=1) for row in atms], axis=1)
            self.plev3 = np.stack([np.stack([a_.level["pressure"] for a_ in row], axis=1) for row in atms], axis=1)
            return
        nfac, nlv = ng * nt, len(prof3["pressure"])
        if self.shared is not None:
            # one wavelength block of a multi-GPU spectrum: the facet set-up and the table rows / weights of every
            # (facet, layer) do not depend on the block -- done once for the grid (setup_facets_3d in _picaso_devices)
            sh = self.shared
            atm_f = copy.copy(sh["atm_f"])
            if sh.get("tall") is not None:
                atm_f._fast_tall = sh["tall"] + (opa,)
            self.atm = _atmosphere_block(sh["atm"], sh["lo"], sh["hi"], wno)
            self.tlev3, self.plev3 = sh["tlev3"], sh["plev3"]
        else:
            atm_f, self.atm, self.tlev3, self.plev3 = setup_facets_3d(inp, opa, wno, ng, nt)
        if self.ngauss > 1:
            # correlated-k tables (justdoit.py:407-421: planes with a trailing ngauss axis): facet-major planes with the
            # Gauss index fastest; the legs solve all nwno*ngauss columns of a facet in one launch (resident.*_3d_ck)
            self.planes3d = optics.compute_opacity_facet_major_ck(atm_f, opa, ng, nt, **co3)
            return
        tabs3 = None
        if (not clear3 and not opt.all_planes and cld3 is not None and inp["test_mode"] is None and not self.full_output
                and not opt.facet_fastest and not opt.host_regrid):
            tabs3 = optics._facet_major_cloud_tables(cld3, nlv - 1, nfac, ctx)   # tables on their own grid, else None
        if (clear3 or tabs3 is not None) and not self.full_output and not opt.facet_fastest:
            # the planes in facet-major layout straight from ONE fused gas + mixing launch over all facets (no cloud:
            # two or three of them; cloud tables on their own grid: interpolated inside that launch); the solvers take
            # every facet as a spectrum of its own (resident.*_3d_fm_batch: same bits)
            self.planes3d = optics.compute_opacity_facet_major(
                atm_f, opa, ng, nt, stream=common["stream"], delta_eddington=common["delta_eddington"],
                raman=common["raman"], exclude_mol=inp["atmosphere"]["exclude_mol"], want=want3, cloud_tables=tabs3)
        else:
            self.planes3d = optics.compute_opacity_facets(atm_f, opa, ng, nt, **co3)

    def _want_1d(self, atm):
        """Which of compute_opacity's 13 planes the legs of a 1-D spectrum read, and under which names.

        Only the planes the requested legs read are written (Toon: 11 for reflected light, 3 for thermal emission, 1 for
        transmission; the SH solvers take the whole set).  ``derive``: planes the reflected kernels re-derive exactly are
        not written at all where the launch can do so (default options; resident.reflected_can_derive): tau, tau_og
        (running sums), gcos2 (0.5 ftau_ray).  ``lean``: cloud-free atmosphere (no cloud profile, no test mode) -- most of
        the 13 planes are exact copies of others or constants (cosb = cosb_og = ftau_cld = 0, ftau_ray = 1, gcos2 = 0.5,
        and with cosb = 0 the delta-scaling is the identity: dtau_og = dtau, tau_og = tau, w0_og = w0), so only dtau, tau
        and w0 are written (0.26 -> 0.09 ms of mixing at 1e5 x 90) and the solvers get the same buffer under several
        names plus three constant planes kept on the opacity object: same values, hence the same bits, as the full set.
        ``sh_lean``: SH4 with the reference's default forms, same atmosphere: dtau and w0 are all the cloud-free SH launch
        reads.  ``sh_top``: a cloud deck -- the layers above it go through the cloud-free SH kernel; a cloudy SH spectrum with
        the default options leaves out the level planes (running products in the kernel).  Correlated-k tables (Toon),
        patchy clouds, test modes and ``all_planes`` take the full set."""
        inp, opt, calc, common, toon = self.inp, self.opt, self.calculation, self.common, self.toon
        plain = (self.ngauss == 1 and inp["test_mode"] is None and not self.do_holes and not opt.all_planes)
        rayleigh = len(getattr(atm, "rayleigh_molecules", [])) > 0
        cloud_free = bool(getattr(atm, "cloud_free", False))
        self.derive = (not self.is_sh and plain and "reflected" in calc and not self.full_output
                       and resident.reflected_can_derive(atm.c.nlevel, self.nwno, self.ng, self.nt, self.ubar0, self.ubar1,
                                                         self.cos_theta, toon["single_phase"], toon["multi_phase"], self.frac[2],
                                                         toon["toon_coefficients"], atm.get_lvl_flux))
        self.lean = not self.is_sh and plain and cloud_free and rayleigh
        self.sh_lean = False
        self.th_w0 = "w0_no_raman"
        want = None
        if self.is_sh:
            sh_o = self.sh
            self.sh_lean = (plain and cloud_free and rayleigh and not self.full_output
                            and resident.reflected_SH_can_derive(
                                common["stream"], sh_o["w_single_form"], sh_o["w_multi_form"], sh_o["psingle_form"],
                                sh_o["w_single_rayleigh"], sh_o["w_multi_rayleigh"], sh_o["psingle_rayleigh"], self.frac[2],
                                sh_o["single_form"], 1 if sh_o["calculate_fluxes"] else 0))
            if self.sh_lean:
                want = {"dtau", "w0"}
            else:
                if inp["test_mode"] is None and rayleigh and not opt.all_planes:
                    # (every wavelength block of a sharded spectrum reads the same profile, hence the same statement)
                    self.sh_top = _cloud_free_top(inp, atm.c.nlayer)
                # the level planes tau / tau_og are running sums: the default-options launch carries the beam exponentials
                # as running products instead of reading them, and cosb, gcos2, w0_no_raman are read by no SH solver
                if (not opt.all_planes and not self.full_output and resident.reflected_SH_can_derive_levels(
                        atm.c.nlevel, self.nwno * self.ngauss, common["stream"], sh_o["w_single_form"], sh_o["w_multi_form"],
                        sh_o["psingle_form"], sh_o["w_single_rayleigh"], sh_o["w_multi_rayleigh"], sh_o["psingle_rayleigh"],
                        self.frac[2], sh_o["single_form"], 1 if (sh_o["calculate_fluxes"] and self.ngauss == 1) else 0)):
                    want = {"dtau", "w0", "cosb_og", "ftau_cld", "ftau_ray", "f_deltaM", "dtau_og", "w0_og"}
        elif self.lean:
            want = set()
            if "reflected" in calc:
                want |= {"dtau", "w0"} if self.derive else {"dtau", "tau", "w0"}
            if "thermal" in calc:
                self.th_w0 = "w0" if (common["raman"] == 2 and "reflected" in calc) else "w0_no_raman"
                want |= {"dtau", self.th_w0}
            if "transmission" in calc:
                want |= {"dtau"}
        else:
            want = set()
            if "reflected" in calc:
                want |= set(resident.REFLECTED_PLANES)
                if self.derive:
                    want -= {"tau", "tau_og", "gcos2"}
            if "thermal" in calc:
                want |= {"dtau_og", "w0_no_raman", "cosb_og"}
            if "transmission" in calc:
                want |= {"dtau_og"}
        if want is not None and not want:            # (None = the whole set, SH) a calculation string that names no leg (full_output of the set-up alone): one plane, no leg reads it
            want = {"dtau"} if self.lean else {"dtau_og"}
        return want

    def _plan_1d(self):
        """justdoit.py:180-252: ATMSETUP, get_opacities, compute_opacity (and the thinned-cloud set of patchy clouds)."""
        inp, opa, wno, shared = self.inp, self.opa, self.wno, self.shared
        if shared is not None:
            self.atm = atm = _atmosphere_block(shared["atm"], shared["lo"], shared["hi"], wno)
        else:
            self.atm = atm = _setup_atmosphere(inp, opa, wno)
        nlayer, nwno, common = atm.c.nlayer, self.nwno, self.common
        if shared is not None and shared.get("plan") is not None:
            opa._plan = shared["plan"]         # table rows / weights per layer: the same for every wavelength block
        else:
            opa.get_opacities(atm, exclude_mol=inp["atmosphere"]["exclude_mol"])
        want = self._want_1d(atm)
        co_kw = dict(ngauss=self.ngauss, stream=common["stream"], delta_eddington=common["delta_eddington"],
                     test_mode=inp["test_mode"], raman=common["raman"], full_output=self.full_output, want=want)
        self.planes = planes = optics.compute_opacity_resident(atm, opa, **co_kw)
        if self.lean and self.derive:
            # the reflected kernel gets dtau and w0 only (everything else re-derived); the thermal one its three names
            zero, _, _ = _constant_planes(opa, nlayer, nwno)
            self.rplanes = {"dtau": planes["dtau"], "w0": planes["w0"]}
            planes.update(dtau_og=planes["dtau"], cosb_og=zero)
            if self.th_w0 == "w0":
                planes["w0_no_raman"] = planes["w0"]
        elif self.sh_lean:
            zero, _, _ = _constant_planes(opa, nlayer, nwno)
            self.rplanes = {"dtau": planes["dtau"], "w0": planes["w0"]}
            # (dtau_og: what the transmission leg reads -- without cloud nothing is delta-scaled, dtau_og IS dtau)
            self.planes = dict(self.rplanes, cosb_og=zero, dtau_og=planes["dtau"])
        elif self.lean:
            zero, one, half = _constant_planes(opa, nlayer, nwno)
            planes.update(dtau_og=planes["dtau"], cosb=zero, cosb_og=zero, ftau_cld=zero, ftau_ray=one, gcos2=half)
            if "tau" in planes:
                planes.update(tau_og=planes["tau"], w0_og=planes["w0"])
            if self.th_w0 == "w0":
                planes["w0_no_raman"] = planes["w0"]
        if self.do_holes:                      # justdoit.py:139-142, 248-252: a second, thinned-cloud column set
            self.planes_clear = optics.compute_opacity_resident(atm, opa, fthin_cld=inp["clouds"]["fthin_cld"],
                                                                do_holes=True, **co_kw)

    # ---------------------------------------------------------------- enqueue
    def enqueue(self):
        """Every leg's kernels, reflected -> thermal -> transmission; the copies back (each a stream synchronisation) and
        the host-side integrals run afterwards (``__call__``), so the GPU goes through the legs back to back while the
        host is still preparing the next launch."""
        enqueued = False
        try:
            with _options.use(self.opt):
                if "reflected" in self.calculation:
                    self._enqueue_reflected()
                if "thermal" in self.calculation:
                    self._enqueue_thermal()
                if "transmission" in self.calculation:
                    self._enqueue_transmission()
            enqueued = True
        finally:
            # The thermal leg may run on a second stream (tctx) that reads the opacity planes of `ctx`.  The planes must
            # not return to ctx's block cache while that stream may still read them: in the normal case the copies back
            # (each a synchronisation of the stream that produced the result) have run by the time they are released --
            # at the end of this call, or, with defer=True, when `finish` (which keeps them alive) is done.  Only an
            # exception in between needs the explicit wait.  (A wait here on every call would also make the blocks of
            # a multi-GPU spectrum, each enqueued with defer=True, take turns instead of running side by side.)
            if self.tctx is not self.ctx and not enqueued:
                try:
                    device.sync(self.tctx)
                except Exception:
                    pass
        # planes read from the second stream stay alive until the results are in; everything else returns to the
        # context's block cache as soon as its kernels are enqueued (reuse is ordered on the stream: the phases of a
        # phase curve recycle one set of plane blocks)
        self.keep_alive = [self.planes, self.planes_clear, self.planes3d] if self.tctx is not self.ctx else []
        return self

    def _blend(self, ctx, solve, out, lvl, compress):
        """The patchy-cloud blend around a solver (justdoit.py:287-305, 346-361): cloudy and thinned-cloud columns
        through the same launch sequence, ``(1 - fhole) cloudy + fhole clear`` on the intensities and on the level fluxes,
        then the disk sum of the blend.  Without holes: one fused launch."""
        if not self.do_holes:
            solve(self.rplanes if (self.rplanes is not None and out is self.xint) else self.planes, out, lvl, True)
            return
        oc = DeviceArray(out.shape, ctx)
        lvc = [DeviceArray(a_.shape, ctx) for a_ in lvl] if lvl else None
        solve(self.planes, out, lvl, False)
        solve(self.planes_clear, oc, lvc, False)
        resident.axpby(ctx, 1.0 - self.fhole, out, self.fhole, oc, out)
        for a_, b_ in zip(lvl or [], lvc or []):
            resident.axpby(ctx, 1.0 - self.fhole, a_, self.fhole, b_, a_)
        compress()

    def _reflected_1d(self, pl, x, lv, fuse):
        """One reflected-light solve of a 1-D column set (justdoit.py:256-307): Toon or SH, the Gauss loop inside."""
        ctx, nlevel, nwno, ng, nt, ngauss = self.ctx, self.nlevel, self.nwno, self.ng, self.nt, self.ngauss
        gw, tw, alb = (self.gweight, self.tweight, self.alb) if fuse else (None, None, None)
        geo = (self.rs, self.ubar0, self.ubar1, self.cos_theta, self.d_f0)
        if self.is_sh:                                           # justdoit.py:259-269
            sh, stream = self.sh, self.common["stream"]
            forms = (sh["w_single_form"], sh["w_multi_form"], sh["psingle_form"], sh["w_single_rayleigh"],
                     sh["w_multi_rayleigh"], sh["psingle_rayleigh"])
            if ngauss > 1:
                resident.reflected_SH_ck(ctx, nlevel, nwno, ngauss, ng, nt, pl, *geo, *forms, *self.tthg[2:], stream,
                                         self.gauss_wts, x, b_top=self.b_top, single_form=sh["single_form"],
                                         cloud_free_above=self.sh_top, gweight=gw, tweight=tw, albedo=alb)
                return
            sh_flux = None                                       # layer moment fluxes, flx = calculate_fluxes
            if sh["calculate_fluxes"]:
                sh_flux = DeviceArray((ng, nt, stream * nlevel, nwno), ctx)
            _reflected_sh(ctx, nlevel, nwno, ng, nt, pl, *geo, sh, *self.tthg[2:], stream, self.b_top, x, self.gweight,
                          self.tweight, self.alb, sh_flux, cloud_free_above=self.sh_top)
            if sh_flux is not None:
                self.atm.flux_layers = sh_flux.to_host()
            return
        toon = self.toon
        if ngauss > 1:
            resident.reflected_1d_ck(ctx, nlevel, nwno, ngauss, ng, nt, pl, *geo, *self.tthg, self.gauss_wts, x,
                                     toon_coefficients=toon["toon_coefficients"], b_top=self.b_top, gweight=gw, tweight=tw,
                                     albedo=alb, lvl_fluxes=lv)
        elif self.batch is not None and lv is None and fuse:
            # spectrum_batch(): the launch is issued later, together with the other spectra's
            self.batch.add_reflected(
                (nlevel, nwno, ng, nt, self.tthg, toon["toon_coefficients"], self.b_top, tuple(self.gweight),
                 tuple(self.tweight), tuple(k_ for k_ in resident.REFLECTED_PLANES if pl.get(k_) is not None)),
                dict(ctx=ctx, planes=pl, rs=self.rs, ubar0=self.ubar0, ubar1=self.ubar1, cos_theta=self.cos_theta,
                     F0PI=self.d_f0, xint=x, albedo=self.alb))
        else:
            _reflected(ctx, nlevel, nwno, ng, nt, pl, *geo, *self.tthg, toon["toon_coefficients"], self.b_top, x, lv,
                       self.gweight, self.tweight, alb)

    def _reflected_3d(self):
        """justdoit.py:488-500: all facets, the Gauss loop inside."""
        ctx, nlevel, nwno, ng, nt, p3 = self.ctx, self.nlevel, self.nwno, self.ng, self.nt, self.planes3d
        geo = (self.rs, self.ubar0, self.ubar1, self.cos_theta, self.d_f0)
        if self.ngauss > 1:
            resident.reflected_3d_ck(ctx, nlevel, nwno, self.ngauss, ng, nt, p3, *geo, *self.tthg, self.gauss_wts, self.xint,
                                     gweight=self.gweight, tweight=self.tweight, albedo=self.alb)
        elif self.batch is not None:                             # phase_curve(): one launch for a chunk of phases
            present = tuple(k for k in resident.REFLECTED_PLANES if p3.get(k) is not None)
            present += ("facet-major",) if p3.get("_fm") else ()
            self.batch.add_reflected_3d((nlevel, nwno, ng, nt, self.tthg, present, tuple(self.gweight), tuple(self.tweight)),
                                        dict(ctx=ctx, planes=p3, rs=self.rs, ubar0=self.ubar0, ubar1=self.ubar1,
                                             cos_theta=self.cos_theta, F0PI=self.d_f0, xint=self.xint, albedo=self.alb))
        else:
            (resident.reflected_3d if not p3.get("_fm") else _reflected_3d_fm)(
                ctx, nlevel, nwno, ng, nt, p3, *geo, *self.tthg, self.xint, self.gweight, self.tweight, self.alb)

    def _enqueue_reflected(self):
        ctx, nwno, ng, nt, atm = self.ctx, self.nwno, self.ng, self.nt, self.atm
        self.xint = xint = DeviceArray((ng, nt, nwno), ctx)
        self.alb_x = DeviceArray((nwno + 1,), ctx)        # [nwno]: the Bond-albedo integral (prefetch)
        self.alb = alb = self.alb_x.head(nwno)
        lvl = None
        if self.dimension == "3d":
            self._reflected_3d()
        else:
            lvl = [DeviceArray((ng, nt, self.nlevel, nwno), ctx) for _ in range(4)] if self.lvl_flux else None
            self._blend(ctx, self._reflected_1d, xint, lvl,
                        lambda: resident.compress_disco(ctx, nwno, self.cos_theta, xint, self.gweight, self.tweight,
                                                        self.d_f0, alb))
        self.dev["albedo"] = alb

        def collect_reflected():          # read back after every leg has been enqueued
            self.returns["albedo"] = _fetch(self.prefetched, "albedo", alb, self.returns, "bond_integral")
            if self.full_output:
                atm.xint_at_top = xint.to_host()
            if lvl is not None:
                # justdoit.py:536-548: every level disk-integrated with compress_disco(..., F0PI = 1):
                # (nlevel, nwno) arrays; on the device over nlevel*nwno columns (F0PI = None means 1)
                atm.lvl_output_reflected = {}
                for key, a_ in zip(("flux_minus", "flux_plus", "flux_minus_mdpt", "flux_plus_mdpt"), lvl):
                    dsum = DeviceArray((self.nlevel, nwno), ctx)
                    resident.compress_disco(ctx, self.nlevel * nwno, self.cos_theta, a_, self.gweight, self.tweight, None, dsum)
                    atm.lvl_output_reflected[key] = dsum.to_host()
        self.collect.append(collect_reflected)

    def _thermal_1d(self, pl, fx, lv, fuse):
        """One thermal solve of a 1-D column set (justdoit.py:328-380): Toon or SH, the Gauss loop inside."""
        tctx, nlevel, nwno, ng, nt, ngauss, atm = self.tctx, self.nlevel, self.nwno, self.ng, self.nt, self.ngauss, self.atm
        tl, pv = atm.level["temperature"], atm.level["pressure"]
        if self.is_sh:                                           # justdoit.py:364-370
            stream, de = self.common["stream"], self.common["delta_eddington"]
            if ngauss > 1:
                # ff = 0 if np.array_equal(cosb, cosb_og) else cosb_og**stream (fluxes.py:3072-3075): see _thermal_sh
                resident.thermal_SH_ck(tctx, nlevel, self.d_wno, nwno, ngauss, ng, nt, tl, pl["dtau"], pl["w0"], pl["cosb_og"],
                                       pv, self.ubar1, self.rs, stream, atm.hard_surface, de, self.gauss_wts, fx,
                                       tau=pl.get("tau"), gweight=self.gweight, tweight=self.tweight, flux_disk=self.disk)
            else:
                _thermal_sh(tctx, nlevel, self.d_wno, nwno, ng, nt, tl, pl, pv, self.ubar1, self.rs, stream,
                            atm.hard_surface, de, fx, self.gweight, self.tweight, self.disk)
            return
        kw = dict(gweight=self.gweight, tweight=self.tweight, flux_disk=self.disk) if fuse else {}
        kw.update(self.tkw)
        if ngauss > 1:
            resident.thermal_1d_ck(tctx, nlevel, self.d_wno, nwno, ngauss, ng, nt, tl, pl["dtau_og"], pl["w0_no_raman"],
                                   pl["cosb_og"], pv, self.ubar1, self.rs, atm.hard_surface, self.gauss_wts, fx,
                                   lvl_fluxes=lv, **kw)
        elif self.batch is not None and lv is None and fuse and not self.tkw:
            self.batch.add_thermal(
                (nlevel, nwno, ng, nt, int(atm.hard_surface), self.d_wno.addr, tuple(self.gweight), tuple(self.tweight)),
                dict(ctx=tctx, wno=self.d_wno, tlevel=np.array(tl, dtype=float), plevel=np.array(pv, dtype=float),
                     dtau=pl["dtau_og"], w0=pl["w0_no_raman"], cosb=pl["cosb_og"], ubar1=self.ubar1, rs=self.rs, flux=fx,
                     disk=self.disk))
        else:
            resident.thermal_1d(tctx, nlevel, self.d_wno, nwno, ng, nt, tl, pl["dtau_og"], pl["w0_no_raman"], pl["cosb_og"],
                                pv, self.ubar1, self.rs, atm.hard_surface, fx, lvl_fluxes=lv, **kw)

    def _thermal_3d(self, flux):
        """justdoit.py:502-516: all facets, the Gauss loop inside."""
        tctx, nlevel, nwno, ng, nt, p3, th3, atm = (self.tctx, self.nlevel, self.nwno, self.ng, self.nt, self.planes3d,
                                                     self.th3, self.atm)
        cosb = p3[th3[2]] if th3[2] else None
        if self.ngauss > 1:
            resident.thermal_3d_ck(tctx, nlevel, self.d_wno, nwno, self.ngauss, ng, nt, self.tlev3, p3[th3[0]], p3[th3[1]],
                                   cosb, self.plev3, self.ubar1, self.rs, atm.hard_surface, self.gauss_wts, flux,
                                   gweight=self.gweight, tweight=self.tweight, flux_disk=self.disk)
        elif self.batch is not None:
            self.batch.add_thermal_3d((nlevel, nwno, ng, nt, int(atm.hard_surface), self.d_wno.addr,
                                       (th3[2] is not None, bool(p3.get("_fm"))), tuple(self.gweight), tuple(self.tweight)),
                                      dict(ctx=self.ctx, wno=self.d_wno, tlevel=np.array(self.tlev3, dtype=float),
                                           plevel=np.array(self.plev3, dtype=float), dtau=p3[th3[0]], w0=p3[th3[1]],
                                           cosb=cosb, ubar1=self.ubar1, rs=self.rs, flux=flux, disk=self.disk, keep=p3))
        elif p3.get("_fm"):
            resident.thermal_3d_fm_batch(tctx, nlevel, self.d_wno, nwno, ng, nt, np.asarray(self.tlev3, dtype=float)[None],
                                         [p3[th3[0]]], [p3[th3[1]]], [cosb] if th3[2] else None,
                                         np.asarray(self.plev3, dtype=float)[None],
                                         np.asarray(self.ubar1, dtype=float).reshape(1, ng, nt), [self.rs], atm.hard_surface,
                                         [flux], self.gweight, self.tweight, [self.disk])
        else:
            resident.thermal_3d(tctx, nlevel, self.d_wno, nwno, ng, nt, self.tlev3, p3[th3[0]], p3[th3[1]], cosb, self.plev3,
                                self.ubar1, self.rs, atm.hard_surface, flux, self.gweight, self.tweight, self.disk)

    def _enqueue_thermal(self):
        tctx, nwno, ng, nt, atm, opa = self.tctx, self.nwno, self.ng, self.nt, self.atm, self.opa
        self.d_wno = _resident_vector(opa, "wno", self.wno, nwno)
        flux = DeviceArray((ng, nt, nwno), tctx)
        self.disk_x = DeviceArray((nwno + 1,), tctx)       # [nwno]: the effective-temperature integral
        self.disk = disk = self.disk_x.head(nwno)
        tlvl = tlvl_disk = None
        if self.dimension == "3d":
            self._thermal_3d(flux)
        else:
            # get_lvl_flux switches the thermal leg to calc_type = 1 with dwno = wno*0 (justdoit.py:322-327,
            # :342): the bin-mean Planck function in wavenumber units, also for the top-of-atmosphere flux
            tlvl = [DeviceArray((ng, nt, self.nlevel, nwno), tctx) for _ in range(4)] if self.lvl_flux else None
            self.tkw = dict(dwno=DeviceArray.zeros((nwno,), tctx), calc_type=1) if tlvl is not None else {}
            self._blend(tctx, self._thermal_1d, flux, tlvl,
                        lambda: resident.compress_thermal(tctx, nwno, flux, self.gweight, self.tweight, disk))
            if tlvl is not None:                              # justdoit.py:575-580, disk sums on the device
                tlvl_disk = []
                for a_ in tlvl:
                    dsum = DeviceArray((self.nlevel, nwno), tctx)
                    resident.compress_thermal(tctx, self.nlevel * nwno, a_, self.gweight, self.tweight, dsum)
                    tlvl_disk.append(dsum)
        self.dev["thermal"] = disk

        def collect_thermal():
            self.returns["thermal"] = _fetch(self.prefetched, "thermal", disk, self.returns, "teff_integral")
            if self.full_output:
                atm.flux_at_top = flux.to_host()
            if tlvl_disk is not None:
                # energy per wavenumber bin: disk-integrated level flux * delta_wno (justdoit.py:575-580)
                delta_wno = getattr(opa, "delta_wno", None)
                if delta_wno is None:
                    delta_wno = np.concatenate((np.diff(self.wno), [np.diff(self.wno)[-1]]))
                atm.lvl_output_thermal = {
                    key: a_.to_host() * delta_wno
                    for key, a_ in zip(("flux_minus", "flux_plus", "flux_minus_mdpt", "flux_plus_mdpt"), tlvl_disk)}
        self.collect.append(collect_thermal)

    def _enqueue_transmission(self):
        """justdoit.py:388-405, :522-523."""
        ctx, nwno, atm = self.ctx, self.nwno, self.atm
        if self.dimension != "1d":
            raise Exception("transmission is a 1-D calculation (the reference has no 3-D branch for it)")
        if self.radius_star == "nostar" or np.isnan(self.radius_star) or np.isnan(atm.planet.radius):
            raise Exception("transmission needs the stellar radius (star()) and the planet radius and "
                            "mass (gravity())")
        tr = DeviceArray((nwno,), ctx)

        def runtr(pl, out):
            resident.transit_1d_ck(ctx, atm.level["z"], atm.level["dz"], self.nlevel, nwno, self.ngauss, self.radius_star,
                                   atm.layer["mmw"], atm.c.k_b, atm.c.amu, atm.level["pressure"],
                                   atm.level["temperature"], atm.layer["colden"], pl["dtau_og"], self.gauss_wts, out)
        runtr(self.planes, tr)
        if self.do_holes:                                     # blend per Gauss point == blend of the sums
            trc = DeviceArray((nwno,), ctx)
            runtr(self.planes_clear, trc)
            resident.axpby(ctx, 1.0 - self.fhole, tr, self.fhole, trc, tr)
        self.dev["transit_depth"] = tr
        self.collect.append(lambda: self.returns.__setitem__("transit_depth", tr.to_host()))

    # ---------------------------------------------------------------- finish
    def prefetch(self, post_ctx=None):
        """The copies of the per-wavelength results go on the stream NOW, behind this spectrum's solver launches, into
        pinned blocks; ``__call__`` then waits for these copies only, while the stream already holds the next spectra's
        launches.  ``post_ctx``: a context whose stream the caller has ordered behind the solvers (``ctx_wait``) -- the
        integrals (four launches that leave the chip empty) and the PCIe copies then run next to the following spectra's
        opacity kernels instead of in front of them.  Preceded by the spectrum-wide integrals of the two results
        (numpy's bits: csrc/integrals.hip), each stored behind its vector so that one copy brings both."""
        opa, wno, nwno = self.opa, self.wno, self.nwno
        whole = not self.raw and nwno > 1 and self.shared is None and not self.opt.host_integrals
        if "albedo" in self.dev and "albedo" not in self.prefetched:
            src, denom = self.alb, None
            if whole:
                d_w, _ = _trapz_resident(opa, wno)
                d_st = self.d_f0 if self.stellar is self.F0PI else _resident_vector(opa, "stellar", self.stellar, nwno)
                denom = _bond_denominator(opa, wno, self.stellar, d_st)
                resident.trapz(post_ctx or self.ctx, nwno, d_w, self.alb, self.alb_x.addr + 8 * nwno, mult=d_st)
                src = self.alb_x
            pc = post_ctx or src.ctx
            self.prefetched["albedo"] = (src.to_host_async(device.PinnedArray(src.shape, pc), pc), denom)
        if "thermal" in self.dev and "thermal" not in self.prefetched:
            src = self.disk
            if whole:
                _, d_wr = _trapz_resident(opa, wno)
                resident.trapz(post_ctx or self.tctx, nwno, d_wr, self.disk, self.disk_x.addr + 8 * nwno, reverse=True)
                src = self.disk_x
            pc = post_ctx or src.ctx
            self.prefetched["thermal"] = (src.to_host_async(device.PinnedArray(src.shape, pc), pc), None)

    def __call__(self):
        """Results are read back leg by leg (each copy waits for the stream that produced it) and a leg's spectrum-wide
        integrals run as soon as it has arrived: the Bond-albedo integral overlaps the thermal kernels still running
        on the second stream.  Same stages, same order of keys as ``_postprocess``."""
        returns, atm = self.returns, self.atm
        out = {"wavenumber": self.wno}
        for fin in self.collect:
            fin()
            if self.raw:
                continue
            if "albedo" in returns and "albedo" not in out:
                _post_reflected(out, returns, self.wno, self.stellar, self.sa, atm.planet.radius, self.opa)
            if "thermal" in returns and "thermal" not in out:
                _post_thermal(out, returns, self.wno, self.stellar, self.radius_star, atm.planet.radius, self.opa)
        del self.keep_alive[:]
        # the collectors are closures over `self`: dropping them breaks the reference cycle, so that this object -- and the
        # device arrays it owns -- go when the caller lets go of it, not when Python's cycle collector next runs (a
        # retrieval loop otherwise piles up hundreds of dead planes between two collections: tools/leak_check.py)
        self.collect = []
        if self.raw:          # one wavelength block of a multi-GPU spectrum: the integrals need the whole grid
            if self.full_output:
                returns["full_output"] = atm.as_dict() if self.as_dict else atm
            return returns
        out = _post_final(out, returns)
        if self.full_output:
            out["full_output"] = atm.as_dict() if self.as_dict else atm
        return out

    def finish(self):
        """The call-by-call path (cloud tables on their own grid, Oklopcic Raman, SH, correlated-k, 3-D): integrals and
        result copies go on the streams behind each leg's kernels, as the C driver does for the plain Toon call."""
        if not self.raw and not self.opt.sync_copies:
            self.prefetch()
        return self()


def _trapz_weights(opa, wno):
    """``diff(1/wno)`` and ``diff(1/wno[::-1])``: the abscissa differences ``np.trapezoid`` forms on every call, kept
    on the opacity object (the grid does not change between the 1e4-1e6 spectra of a retrieval; at 1e5 wavelengths
    the three integrals of a reflected + thermal spectrum were 0.4 ms of the 1.5 ms call)."""
    hit = opa.__dict__.get("_trapz")
    if hit is None or hit[0] is not wno:
        inv = 1 / wno
        hit = (wno, np.diff(inv), np.diff(inv[::-1]))
        opa.__dict__["_trapz"] = hit
    return hit[1], hit[2]


def _trapz(d, y, buf=None):
    """``np.trapezoid(y, x)`` with ``d = diff(x)`` given: numpy's own expression, so the same bits.  ``buf``: a scratch
    array of ``d``'s shape for the intermediate results (three fresh 0.8 MB arrays per integral at 1e5 wavelengths
    cost more than the arithmetic)."""
    if buf is None or buf.shape != d.shape:
        return (d * (y[1:] + y[:-1]) / 2.0).sum(-1)
    np.add(y[1:], y[:-1], out=buf)
    np.multiply(d, buf, out=buf)
    np.divide(buf, 2.0, out=buf)
    return buf.sum(-1)


def _trapz_scratch(opa, n):
    """Two scratch vectors kept on the opacity object for the spectrum-wide integrals."""
    hit = opa.__dict__.get("_trapz_buf")
    if hit is None or hit[0].shape != (n - 1,):
        hit = (np.empty(n - 1), np.empty(n))
        opa.__dict__["_trapz_buf"] = hit
    return hit


def _trapz_resident(opa, wno):
    """``_trapz_weights`` in HBM (``picaso_trapz_dev``), uploaded once per grid."""
    d, dr = _trapz_weights(opa, wno)
    hit = opa.__dict__.get("_trapz_dev")
    if hit is None or hit[0] is not d:
        hit = (d, DeviceArray.from_host(d, opa.ctx), DeviceArray.from_host(dr, opa.ctx))
        opa.__dict__["_trapz_dev"] = hit
    return hit[1], hit[2]


def _bond_denominator(opa, wno, stellar, d_stellar):
    """``np.trapz(x=1/wno, y=stellar)``, kept while the resident copy of the stellar spectrum is the same object
    (``_resident_vector`` replaces it when the content changes)."""
    hit = opa.__dict__.get("_bond_denom")
    if hit is None or hit[0] is not d_stellar or hit[1] is not wno:
        d, _ = _trapz_weights(opa, wno)
        hit = (d_stellar, wno, _trapz(d, np.zeros(len(wno)) + np.asarray(stellar, dtype=float)))
        opa.__dict__["_bond_denom"] = hit
    return hit[2]


def _post_reflected(out, raw, wno, stellar, sa, planet_radius, opa=None):
    """Bond albedo (Batalha+2019 eq. 18) and the reflected planet-to-star flux ratio (justdoit.py:552-566).
    ``raw["bond_integral"]`` = (numerator integrated on the device, denominator) when the caller had them."""
    albedo = raw["albedo"]
    out["albedo"] = albedo
    if raw.get("bond_integral") is not None:
        num, denom = raw["bond_integral"]
        out["bond_albedo"] = num / denom
    elif opa is not None:
        d, _ = _trapz_weights(opa, wno)
        b1, b2 = _trapz_scratch(opa, len(wno))
        # the denominator does not change while the stellar spectrum does not: kept with a copy it is compared against
        # (one pass instead of the integral's four; a read-only array -- the no-star ones -- is known by identity)
        hit = opa.__dict__.get("_bond_denom_host")
        if (hit is not None and hit[0] is stellar and hit[1] is wno and isinstance(stellar, np.ndarray)
                and (not stellar.flags.writeable or np.array_equal(stellar, hit[2]))):
            denom = hit[3]
        else:
            denom = _trapz(d, stellar, b1)
            if isinstance(stellar, np.ndarray):
                opa.__dict__["_bond_denom_host"] = (stellar, wno, None if not stellar.flags.writeable else stellar.copy(),
                                                    denom)
        np.multiply(albedo, stellar, out=b2)
        out["bond_albedo"] = _trapz(d, b2, b1) / denom
    else:
        out["bond_albedo"] = (np.trapezoid(x=1 / wno, y=albedo * stellar) / np.trapezoid(x=1 / wno, y=stellar))
    if (not np.isnan(sa)) and (not np.isnan(planet_radius)):
        out["fpfs_reflected"] = albedo * (planet_radius / sa) ** 2.0
    else:
        out["fpfs_reflected"] = []


def _post_thermal(out, raw, wno, stellar, radius_star, planet_radius, opa=None):
    """Effective temperature and the thermal planet-to-star flux ratio (justdoit.py:567-599)."""
    thermal = raw["thermal"]
    out["thermal"] = thermal
    out["thermal_unit"] = "erg/s/(cm^2)/(cm)"
    if raw.get("teff_integral") is not None:
        out["effective_temperature"] = (raw["teff_integral"] / 5.67e-5) ** 0.25
    elif opa is not None:
        _, dr = _trapz_weights(opa, wno)
        b1, _ = _trapz_scratch(opa, len(wno))
        out["effective_temperature"] = (_trapz(dr, thermal[::-1], b1) / 5.67e-5) ** 0.25
    else:
        out["effective_temperature"] = (np.trapezoid(x=1 / wno[::-1], y=thermal[::-1]) / 5.67e-5) ** 0.25
    if radius_star == "nostar":
        out["fpfs_thermal"] = ["No star mode for Brown Dwarfs was used"]
    elif (not np.isnan(planet_radius)) and (not np.isnan(radius_star)):
        out["fpfs_thermal"] = thermal / stellar * (planet_radius / radius_star) ** 2.0
    else:
        out["fpfs_thermal"] = []


def _post_final(out, raw):
    if "transit_depth" in raw:
        out["transit_depth"] = raw["transit_depth"]
    if ("fpfs_reflected" in out) and ("fpfs_thermal" in out):
        if (not isinstance(out["fpfs_reflected"], list)) and (not isinstance(out["fpfs_thermal"], list)):
            out["fpfs_total"] = out["fpfs_thermal"] + out["fpfs_reflected"]
    return out


def _postprocess(raw, wno, stellar, sa, radius_star, planet_radius, opa=None):
    """The spectrum-wide quantities of the reference's return dictionary (justdoit.py:552-599) from the
    per-wavelength results: Bond albedo, planet-to-star flux ratios, effective temperature.  Separate from the
    solve so that a spectrum computed in wavelength blocks on several GPUs goes through exactly the same arithmetic
    on the gathered arrays as a single-GPU one (which runs the three stages as its results arrive, see ``picaso``)."""
    out = {"wavenumber": wno}
    if "albedo" in raw:
        _post_reflected(out, raw, wno, stellar, sa, planet_radius, opa)
    if "thermal" in raw:
        _post_thermal(out, raw, wno, stellar, radius_star, planet_radius, opa)
    return _post_final(out, raw)


def _reflected(ctx, nlevel, nwno, ng, nt, planes, rs, ubar0, ubar1, cos_theta, F0PI, single_phase,
               multi_phase, frac_a, frac_b, frac_c, constant_back, constant_forward,
               toon_coefficients, b_top, xint, lvl, gweight, tweight, albedo):
    import ctypes
    from ._lib import check, f64, load, ptr
    u0, u1 = f64(ubar0, (ng, nt)), f64(ubar1, (ng, nt))
    gw, tw = f64(gweight), f64(tweight)
    ci, cd = ctypes.c_int, ctypes.c_double
    check(load().picaso_get_reflected_1d_dev(
        ctx, ci(nlevel), ci(nwno), ctypes.c_long(nwno), ci(ng), ci(nt),
        *[ptr(planes[k].addr) if planes.get(k) is not None else None for k in resident.REFLECTED_PLANES],
        ptr(rs.addr), ptr(u0), ptr(u1),
        cd(cos_theta), ptr(F0PI.addr), ci(single_phase), ci(multi_phase), cd(frac_a), cd(frac_b),
        cd(frac_c), cd(constant_back), cd(constant_forward), ci(1), ci(1 if lvl else 0),
        ci(toon_coefficients), cd(b_top), ptr(xint.addr),
        *[ptr(l.addr) if lvl else None for l in (lvl or [None] * 4)], ptr(gw), ptr(tw),
        ptr(albedo.addr) if albedo is not None else None), ctx)


def _reflected_sh(ctx, nlevel, nwno, ng, nt, planes, rs, ubar0, ubar1, cos_theta, F0PI, sh, frac_a,
                  frac_b, frac_c, constant_back, constant_forward, stream, b_top, xint, gweight,
                  tweight, albedo, flux=None, cloud_free_above=0):
    import ctypes
    from ._lib import check, f64, load, ptr
    u0, u1 = f64(ubar0, (ng, nt)), f64(ubar1, (ng, nt))
    gw, tw = f64(gweight), f64(tweight)
    ci, cd = ctypes.c_int, ctypes.c_double
    names = ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray", "f_deltaM", "dtau_og", "tau_og",
             "w0_og", "cosb_og")
    check(load().picaso_get_reflected_SH_top_dev(
        ctx, ci(nlevel), ci(nwno), ctypes.c_long(nwno), ci(ng), ci(nt),
        *[ptr(planes[k].addr) if planes.get(k) is not None else None for k in names], ptr(rs.addr), ptr(u0), ptr(u1),
        cd(cos_theta),
        ptr(F0PI.addr), ci(sh["w_single_form"]), ci(sh["w_multi_form"]), ci(sh["psingle_form"]),
        ci(sh["w_single_rayleigh"]), ci(sh["w_multi_rayleigh"]), ci(sh["psingle_rayleigh"]),
        cd(frac_a), cd(frac_b), cd(frac_c), cd(constant_back), cd(constant_forward), ci(stream),
        cd(b_top), ci(1 if flux is not None else 0), ci(sh["single_form"]), ci(1), ci(int(cloud_free_above)),
        ptr(xint.addr),
        ptr(flux.addr) if flux is not None else None, ptr(gw), ptr(tw),
        ptr(albedo.addr)), ctx)


def _thermal_sh(ctx, nlevel, d_wno, nwno, ng, nt, tlevel, planes, plevel, ubar1, rs, stream,
                hard_surface, delta_eddington, flux, gweight, tweight, disk):
    import ctypes
    from ._lib import check, f64, load, ptr
    u1 = f64(ubar1, (ng, nt))
    gw, tw = f64(gweight), f64(tweight)
    tl, pl = f64(tlevel), f64(plevel)
    ci = ctypes.c_int
    # ff = 0 if np.array_equal(cosb, cosb_og) else cosb_og**stream (fluxes.py:3072-3075).  Without delta-Eddington
    # scaling the two planes are the same array; with it they are equal only where cosb_og**stream vanishes against
    # cosb_og, and there `cosb_og**stream` IS the reference's 0 (exactly for a cloud-free atmosphere, to < 1e-21 in the
    # weights otherwise): the kernel forms it per element, and nothing is copied back to decide (a 72 MB read of
    # f_deltaM per call used to sit here: 9.5 of the 11.7 ms of an SH4 spectrum at 1e5 wavelengths).
    differs = 1 if delta_eddington else 0
    check(load().picaso_get_thermal_SH_dev(
        ctx, ci(nlevel), ptr(d_wno.addr), ci(nwno), ctypes.c_long(nwno), ci(ng), ci(nt), ptr(tl),
        ptr(planes["dtau"].addr), ptr(planes["tau"].addr) if planes.get("tau") is not None else None,   # tau: never read
        ptr(planes["w0"].addr),
        ptr(planes["cosb_og"].addr), ptr(pl), ptr(u1), ptr(rs.addr), ci(stream), ci(int(hard_surface)),
        ci(differs), ci(0), ptr(flux.addr), ptr(gw), ptr(tw), ptr(disk.addr)), ctx)
