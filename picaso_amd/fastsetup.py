"""One C call for the host-side set-up of a 1-D spectrum (``picaso_host_setup``, ``csrc/setup.hip``): the level / layer
state of ``ATMSETUP`` (reference atmsetup.py:74-461), the table rows and weights of ``get_opacities`` (optics.py:2048-2123,
2241-2306) and the per-layer coefficients of the TAUGAS / TAURAY sums (optics.py:144-277).

``setup(inp, opa, wno)`` returns an ``ATMSETUP`` filled exactly as ``justdoit._setup_atmosphere`` + ``opa.get_opacities``
+ ``optics._layer_factors`` would fill it (``tests/test_fast_setup.py``: every array ``np.array_equal``), or ``None`` when
the call is outside the C function's scope -- an ``e-`` column, ``H-`` / ``H2-``
continua, nearest-neighbour tables, k-tables mixed on the fly, ``exclude_mol`` -- and the caller takes the numpy
mirror.  Premixed correlated-k tables (``RetrieveCKs``) are in scope from round 5: the same ragged-grid search with the
k-table's row numbering, the bracketing continuum temperatures with their 1/T weight, ``mol_fac = colden / mmw``.  ``PICASO_AMD_PY_SETUP=1`` forces the mirror (A/B)."""
import ctypes

import numpy as np

from .options import current as _options

from . import _lib
from .atmsetup import ATMSETUP, molecular_weight

_vp, _ci, _cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double


class SetupArgs(ctypes.Structure):
    _fields_ = [("nlevel", _ci), ("nmol", _ci), ("pressure_bar", _vp), ("temperature", _vp), ("mix", _vp), ("weights", _vp),
                ("gravity", _cd), ("radius", _cd), ("GM", _cd), ("p_reference_bar", _cd), ("pconv", _cd), ("k_b", _cd), ("amu", _cd),
                ("coef1_scale", _cd), ("coef1_den", _cd), ("log_pratio", _vp), ("log10_player", _vp), ("pbar_cubed_hi", _vp), ("pbar_cubed_lo", _vp),
                ("nt", _ci), ("npg", _ci), ("t_inv_grid", _vp), ("p_log_grid", _vp), ("nc_p", _vp), ("row_lut", _vp),
                ("nlut", _ci), ("ncia_t", _ci), ("cia_temps", _vp), ("nopa", _ci), ("ncont", _ci), ("nray", _ci),
                ("opa_idx", _vp), ("cont_a", _vp), ("cont_b", _vp), ("ray_idx", _vp),
                ("level_pressure", _vp), ("level_mmw", _vp), ("level_den", _vp), ("z", _vp), ("dz", _vp),
                ("scale_height", _vp), ("layer_temperature", _vp), ("layer_pressure", _vp), ("layer_mmw", _vp),
                ("layer_gravity", _vp), ("colden", _vp), ("layer_mix", _vp), ("rows", _vp), ("wts", _vp), ("cia_rows", _vp),
                ("mol_fac", _vp), ("cont_fac", _vp), ("ray_fac", _vp), ("pt_opa_index", _vp), ("n_pt_opa_index", _vp),
                ("scratch", _vp), ("premixed", _ci), ("cont_interp", _ci), ("cia_rows2", _vp), ("cia_wts2", _vp)]


def _addr(a):
    return a.__array_interface__["data"][0]


def _is_premixed(opa):
    """A RetrieveCKs object with a premixed table loaded (not mixing on the fly)."""
    return getattr(opa, "_kappa", None) is not None and not getattr(opa, "on_fly", False) and hasattr(opa, "temps")


def _in_scope(opa):
    """Opacity objects the C set-up covers: monochromatic tables with 'linear' interpolation, premixed k-tables."""
    if _is_premixed(opa):
        return True
    return (getattr(opa, "query_method", None) == "linear" and getattr(opa, "ngauss", 1) == 1 and hasattr(opa, "_row_lut")
            and not getattr(opa, "on_fly", False))


class _Signature:
    """What depends on the profile's column names and the opacity object only: molecule lists, index tables, the
    constant half of the argument struct."""

    def __init__(self, cols, opa):
        self.ok = False
        weights, molecules = {}, []
        for k in cols:
            if k in ("pressure", "temperature"):
                continue
            if k == "e-":
                return                                   # electrons: H- / H2- continua, the mirror's case
            try:
                weights[k] = molecular_weight(k)
            except KeyError:
                return                                   # unrecognised column: the mirror warns and skips it
            molecules.append(k)
        if not molecules or "pressure" not in cols or "temperature" not in cols:
            return
        self.all_molecules, self.weights_dict = molecules, weights
        self.weights = np.array([weights[m] for m in molecules], dtype=np.float64)
        # get_needed_continuum (atmsetup.py:248-283) on the full list, then the molecules without line opacities dropped
        avail = opa.avail_continuum
        self.continuum_molecules = [[m1, m2] for m1 in molecules for m2 in molecules if m1 + m2 in avail]
        if ("H-" in molecules and "H-bf" in avail):
            return
        self.rayleigh_molecules = [m for m in molecules if m in opa.rayleigh_molecules]
        # premixed correlated-k tables (RetrieveCKs with a table loaded, not mixing on the fly): no per-molecule line
        # opacities -- ONE "molecule", the premixed table, whose rows are p * ntemp + t of the same ragged-grid search
        self.premixed = _is_premixed(opa)
        opam = set(opa.molecules)
        self.no_opa = [m for m in molecules if m not in opam]
        self.molecules = [m for m in molecules if m in opam]
        self.nopa = 1 if self.premixed else len(self.molecules)
        self.cia_pairs = [a + b for a, b in self.continuum_molecules]
        self.ray_names = [m for m in self.rayleigh_molecules if m in opa._ray]
        ix = {m: i for i, m in enumerate(molecules)}
        i32 = lambda xs: np.ascontiguousarray(xs, dtype=np.int32)
        self.opa_idx, self.ray_idx = i32([ix[m] for m in self.molecules]), i32([ix[m] for m in self.ray_names])
        self.cont_a, self.cont_b = i32([ix[a] for a, _ in self.continuum_molecules]), i32([ix[b] for _, b in self.continuum_molecules])
        if self.premixed:
            # get_mixing_indices' grids (optics.py:1200-1278)
            self.t_inv_grid = np.ascontiguousarray(1 / np.asarray(opa.temps, dtype=np.float64))
            self.p_log_grid = np.ascontiguousarray(np.log10(opa.pressures[opa.pressures > 0]), dtype=np.float64)
            self.nc_p = np.ascontiguousarray(opa.nc_p, dtype=np.int64)
            if self.nc_p.size != self.t_inv_grid.size:
                return
            self.row_lut = np.zeros(1, dtype=np.int64)         # not read: rows are p * ntemp + t
        else:
            self.t_inv_grid = np.ascontiguousarray(opa.t_inv_grid, dtype=np.float64)
            self.p_log_grid = np.ascontiguousarray(opa.p_log_grid, dtype=np.float64)
            self.nc_p = np.ascontiguousarray(opa.nc_p, dtype=np.int64)
            self.row_lut = np.ascontiguousarray(opa._row_lut, dtype=np.int64)
        self.cia_temps = np.ascontiguousarray(np.unique(opa.cia_temps), dtype=np.float64)
        if self.cia_temps.size < 1 or self.t_inv_grid.size < 2:
            return
        self.layouts = {}
        self.ok = True


class _Layout:
    """Output buffers of one (signature, level count): sizes and offsets of the carved arrays, and the argument struct
    with everything that does not change between calls filled in."""
    F_NAMES = ("level_pressure", "level_mmw", "level_den", "z", "dz", "scale_height", "layer_temperature", "layer_pressure",
               "layer_mmw", "layer_gravity", "colden", "layer_mix", "wts", "mol_fac", "cont_fac", "ray_fac", "scratch")

    def __init__(self, sig, n, c):
        nl = n - 1
        nmol = len(sig.all_molecules)
        self.nopa, self.ncont, self.nray = sig.nopa, len(sig.continuum_molecules), len(sig.ray_names)
        nopa, ncont, nray = self.nopa, self.ncont, self.nray
        self.sizes = [n] * 6 + [nl] * 5 + [nmol * nl, nopa * nl * 4, nopa * nl, ncont * nl, nray * nl, 3 * n]
        self.offs = np.concatenate(([0], np.cumsum(self.sizes)))[:-1].tolist()
        self.nf = sum(self.sizes)
        self.o_cia, self.o_pt = nopa * nl * 4, nopa * nl * 4 + max(ncont, 1) * nl
        self.ni = self.o_pt + 4 * nl + 1
        self.f_fields = [(k, 8 * o) for k, o in zip(self.F_NAMES, self.offs)]
        self.i_fields = [("rows", 0), ("cia_rows", 4 * self.o_cia), ("pt_opa_index", 4 * self.o_pt),
                         ("n_pt_opa_index", 4 * (self.o_pt + 4 * nl))]
        a = SetupArgs()
        a.nlevel, a.nmol, a.weights = n, nmol, _addr(sig.weights)
        a.pconv, a.k_b, a.amu = c.pconv, c.k_b, c.amu
        a.coef1_scale = c.rgas * 273.15 ** 2 * .5E5
        a.nt, a.npg = sig.t_inv_grid.size, sig.p_log_grid.size
        a.t_inv_grid, a.p_log_grid, a.nc_p, a.row_lut = (_addr(sig.t_inv_grid), _addr(sig.p_log_grid), _addr(sig.nc_p),
                                                         _addr(sig.row_lut))
        a.nlut, a.ncia_t, a.cia_temps = sig.row_lut.size, sig.cia_temps.size, _addr(sig.cia_temps)
        a.nopa, a.ncont, a.nray = nopa, ncont, nray
        a.opa_idx, a.cont_a, a.cont_b, a.ray_idx = _addr(sig.opa_idx), _addr(sig.cont_a), _addr(sig.cont_b), _addr(sig.ray_idx)
        a.premixed = a.cont_interp = 1 if sig.premixed else 0
        self.template = bytes(a)


class _PressureGrid:
    """The three transcendental arrays, from numpy, for one pressure grid."""
    addr = None

    def __init__(self, pbar, pconv):
        self.pbar = pbar.copy()
        P = pbar * pconv
        self.log_pratio = np.ascontiguousarray(np.log(P[1:] / P[:-1]))
        self.log10_player = np.ascontiguousarray(np.log10(np.sqrt(P[1:] * P[:-1]) / pconv))
        pb = P / pconv
        self.cube_hi, self.cube_lo = np.ascontiguousarray(pb[1:] ** 3), np.ascontiguousarray(pb[:-1] ** 3)


def setup(inp, opa, wno):
    if _options().py_setup:
        return None
    if not _in_scope(opa):
        return None
    at = inp["atmosphere"]
    if at.get("exclude_mol", 1) != 1:
        return None
    radius, mass = inp["planet"]["radius"], inp["planet"]["mass"]
    if not isinstance(radius, float) or (radius == radius and not isinstance(mass, float)):
        return None
    read = at["profile"]
    if read is None or not hasattr(read, "keys"):
        return None
    cols = tuple(read.keys())
    cache = opa.__dict__.setdefault("_fast_setup", {})
    sig = cache.get(cols)
    if sig is None:
        if len(cache) > 16:
            cache.clear()
        sig = cache[cols] = _Signature(cols, opa)
    if not sig.ok:
        return None
    pbar = np.ascontiguousarray(read["pressure"], dtype=np.float64)
    T = np.ascontiguousarray(read["temperature"], dtype=np.float64)
    if pbar.ndim != 1 or T.shape != pbar.shape or pbar.size < 2:
        return None
    n = pbar.size
    nl = n - 1
    mixcols = [np.ascontiguousarray(read[m], dtype=np.float64) for m in sig.all_molecules]
    for c in mixcols:
        if c.shape != pbar.shape:
            return None
    atm = ATMSETUP(inp)
    c = atm.c
    pg = cache.get("pressure")
    if pg is None or pg.pbar.shape != pbar.shape or not np.array_equal(pg.pbar, pbar):
        pg = cache["pressure"] = _PressureGrid(pbar, c.pconv)
    nmol = len(mixcols)
    lay = sig.layouts.get(n)
    if lay is None:
        lay = sig.layouts[n] = _Layout(sig, n, c)
    nopa, ncont, nray, sizes, offs, o_cia, o_pt = lay.nopa, lay.ncont, lay.nray, lay.sizes, lay.offs, lay.o_cia, lay.o_pt
    fbuf = np.empty(lay.nf)
    ibuf = np.empty(lay.ni, dtype=np.int32)
    fb, ib = _addr(fbuf), _addr(ibuf)
    mixp = (_vp * nmol)(*[_addr(x) for x in mixcols])
    gravity = inp["planet"]["gravity"]
    a = SetupArgs.from_buffer_copy(lay.template)            # the grid / index half is the same for every call
    a.pressure_bar, a.temperature, a.mix = _addr(pbar), _addr(T), ctypes.addressof(mixp)
    a.gravity, a.radius, a.p_reference_bar = float(gravity), radius, float(inp["approx"]["p_reference"])
    a.GM = c.G * mass if radius == radius else 0.0
    a.coef1_den = 1.01325 ** 2 * (gravity / 100.0)
    if pg.addr is None:
        pg.addr = (_addr(pg.log_pratio), _addr(pg.log10_player), _addr(pg.cube_hi), _addr(pg.cube_lo))
    a.log_pratio, a.log10_player, a.pbar_cubed_hi, a.pbar_cubed_lo = pg.addr
    for name, off in lay.f_fields:
        setattr(a, name, fb + off)
    for name, off in lay.i_fields:
        setattr(a, name, ib + off)
    rows2 = wts2 = None
    if sig.premixed:
        rows2, wts2 = np.empty((nl, 2), dtype=np.int32), np.empty((nl, 2))
        a.cia_rows2, a.cia_wts2 = _addr(rows2), _addr(wts2)
    rc = _lib.load().picaso_host_setup(ctypes.byref(a))
    if rc != 0:
        return None

    def f(k, shape):
        return fbuf[offs[k]:offs[k] + sizes[k]].reshape(shape)
    # ---- the ATMSETUP the mirror would have built (justdoit._setup_atmosphere) ----
    atm.surf_reflect = inp.get("surface_reflect", 0)
    atm.hard_surface = inp.get("hard_surface", 0)
    atm.wavenumber = wno
    atm.planet.gravity, atm.planet.radius, atm.planet.mass = gravity, radius, inp["planet"]["mass"]
    atm.get_lvl_flux = inp["approx"].get("get_lvl_flux", False)
    atm.weights = sig.weights_dict
    lmix = f(11, (nmol, nl))
    atm.level.update(mixingratios=dict(zip(sig.all_molecules, mixcols)), temperature=T, pressure_bar=pbar,
                     pressure=f(0, (n,)), mmw=f(1, (n,)), den=f(2, (n,)), z=f(3, (n,)), dz=f(4, (n,)), scale_height=f(5, (n,)))
    atm.layer.update(mixingratios={m: lmix[i] for i, m in enumerate(sig.all_molecules)}, temperature=f(6, (nl,)),
                     pressure=f(7, (nl,)), mmw=f(8, (nl,)), gravity=f(9, (nl,)), colden=f(10, (nl,)))
    if not sig.premixed:
        npt = int(ibuf[o_pt + 4 * nl])
        atm.layer["pt_opa_index"] = ibuf[o_pt:o_pt + npt].astype(np.int64)
    c.nlevel, c.nlayer = n, nl
    atm.continuum_molecules = [list(p) for p in sig.continuum_molecules]
    atm.rayleigh_molecules = list(sig.rayleigh_molecules)
    atm.get_clouds(wno)
    if sig.no_opa:
        atm.add_warnings("I found chemistry for these but I do not have computed individual line "
                         "opacities (not including continuum) for: " + ",".join(sig.no_opa))
    atm.molecules = np.array(sig.molecules)
    cia_rows = ibuf[o_cia:o_cia + nl]
    if sig.premixed:                  # RetrieveCKs.get_opacities' plan (premixed table rows, bracketing continuum temperatures)
        plan = dict(premixed=True, molecules=["premixed"], rows=ibuf[:nl * 4].reshape(1, nl, 4), wts=f(12, (1, nl, 4)),
                    fac=np.ones(1), nlayer=nl, cia_pairs=list(sig.cia_pairs), cia_rows=rows2, cia_wts=wts2)
    else:
        plan = dict(molecules=list(sig.molecules), rows=ibuf[:nopa * nl * 4].reshape(nopa, nl, 4), wts=f(12, (nopa, nl, 4)),
                    fac=np.ones(nopa), cia_pairs=list(sig.cia_pairs), cia_rows=cia_rows, nlayer=nl)
    factors = (f(13, (nopa, nl)), f(14, (ncont, nl)), list(sig.ray_names), f(15, (nray, nl)))
    plan["_factors"] = (atm.layer["mixingratios"], factors)
    atm._fast = (plan, factors, opa, (fbuf, ibuf, mixp))
    return atm


def setup_facets(inp, opa, wno, prof_f):
    """The facet-form ATMSETUP of the 3-D path (``justdoit.picaso``: columns ``(nlevel, nfacets)`` for what depends on the
    facet, ``(nlevel, 1)`` for what does not; the pressure grid is shared) from ``picaso_host_setup_facets``, with the tall
    plan and per-layer coefficients ``optics.gas_stage_facets`` would form (facet-major ``nfacets * nlayer`` layers) attached
    as ``atm._fast_tall``.  ``None`` outside the C function's scope."""
    if _options().py_setup:
        return None
    if not _in_scope(opa) or inp["atmosphere"].get("exclude_mol", 1) != 1:
        return None
    radius, mass = inp["planet"]["radius"], inp["planet"]["mass"]
    if not isinstance(radius, float) or (radius == radius and not isinstance(mass, float)):
        return None
    cols = tuple(prof_f.keys())
    cache = opa.__dict__.setdefault("_fast_setup", {})
    sig = cache.get(cols)
    if sig is None:
        if len(cache) > 16:
            cache.clear()
        sig = cache[cols] = _Signature(cols, opa)
    if not sig.ok:
        return None
    pcol = np.asarray(prof_f["pressure"], dtype=np.float64)
    tcol = np.asarray(prof_f["temperature"], dtype=np.float64)
    if pcol.ndim != 2 or pcol.shape[1] != 1 or tcol.ndim != 2 or tcol.shape[0] != pcol.shape[0]:
        return None
    n, nfac = tcol.shape
    nl = n - 1
    if n < 2 or nfac < 1:
        return None
    pbar = np.ascontiguousarray(pcol[:, 0])
    T = np.ascontiguousarray(tcol.T)                                  # (nfacets, nlevel)
    mixcols, strides, shared = [], [], []
    for m in sig.all_molecules:
        v = np.asarray(prof_f[m], dtype=np.float64)
        if v.ndim != 2 or v.shape[0] != n or v.shape[1] not in (1, nfac):
            return None
        mixcols.append(np.ascontiguousarray(v.T if v.shape[1] > 1 else v[:, 0]))
        strides.append(0 if v.shape[1] == 1 else n)
        shared.append(v.shape[1] == 1)
    atm = ATMSETUP(dict(inp, atmosphere=dict(inp["atmosphere"], profile=prof_f), clouds=dict(inp["clouds"], profile=None)))
    c = atm.c
    pg = cache.get("pressure")
    if pg is None or pg.pbar.shape != pbar.shape or not np.array_equal(pg.pbar, pbar):
        pg = cache["pressure"] = _PressureGrid(pbar, c.pconv)
    nmol = len(mixcols)
    lay = sig.layouts.get(n)
    if lay is None:
        lay = sig.layouts[n] = _Layout(sig, n, c)
    nopa, ncont, nray, sizes = lay.nopa, lay.ncont, lay.nray, lay.sizes
    nc1 = max(ncont, 1)
    fsz = sizes[:-1]                                                  # per-facet float outputs (scratch is shared)
    foff = np.concatenate(([0], np.cumsum([x * nfac for x in fsz]))).tolist()
    fbuf = np.empty(foff[-1] + sizes[-1])
    isz = [nopa * nl * 4, nc1 * nl, 4 * nl, 1]
    ioff = np.concatenate(([0], np.cumsum([x * nfac for x in isz]))).tolist()
    ibuf = np.empty(ioff[-1], dtype=np.int32)
    fb, ib = _addr(fbuf), _addr(ibuf)
    mixp = (_vp * nmol)(*[_addr(x) for x in mixcols])
    mstr = (ctypes.c_long * nmol)(*strides)
    gravity = inp["planet"]["gravity"]
    a = SetupArgs.from_buffer_copy(lay.template)
    a.pressure_bar, a.temperature, a.mix = _addr(pbar), _addr(T), ctypes.addressof(mixp)
    a.gravity, a.radius, a.p_reference_bar = float(gravity), radius, float(inp["approx"]["p_reference"])
    # a planet radius: gravity G M / z^2 level by level with libm's pow, facet by facet -- what the mirror's facet form does
    # element by element (atmsetup.get_altitude: math.pow per facet) and the reference's per-facet ATMSETUP with its scalars
    a.GM = c.G * mass if radius == radius else 0.0
    a.coef1_den = 1.01325 ** 2 * (gravity / 100.0)
    if pg.addr is None:
        pg.addr = (_addr(pg.log_pratio), _addr(pg.log10_player), _addr(pg.cube_hi), _addr(pg.cube_lo))
    a.log_pratio, a.log10_player, a.pbar_cubed_hi, a.pbar_cubed_lo = pg.addr
    for k, name in enumerate(_Layout.F_NAMES):
        setattr(a, name, fb + 8 * foff[k])
    for name, o in zip(("rows", "cia_rows", "pt_opa_index", "n_pt_opa_index"), ioff):
        setattr(a, name, ib + 4 * o)
    rows2 = wts2 = None
    if sig.premixed:
        rows2, wts2 = np.empty((nfac * nl, 2), dtype=np.int32), np.empty((nfac * nl, 2))
        a.cia_rows2, a.cia_wts2 = _addr(rows2), _addr(wts2)
    rc = _lib.load().picaso_host_setup_facets(ctypes.byref(a), _ci(nfac), ctypes.c_long(n), mstr)
    if rc != 0:
        return None

    def f(k, shape):                                                  # facet-major output k as (nfacets,) + shape
        return fbuf[foff[k]:foff[k + 1]].reshape((nfac,) + shape)

    def facet_form(x, is_shared):                                     # (nfacets, rows) -> (rows, nfacets | 1) as the mirror holds it
        return np.ascontiguousarray(x[:1].T) if is_shared else np.ascontiguousarray(x.T)
    all_shared = all(shared)
    atm.surf_reflect = inp.get("surface_reflect", 0)
    atm.hard_surface = inp.get("hard_surface", 0)
    atm.wavenumber = wno
    atm.planet.gravity, atm.planet.radius, atm.planet.mass = gravity, radius, mass
    atm.get_lvl_flux = inp["approx"].get("get_lvl_flux", False)
    atm.weights = sig.weights_dict
    lmix = f(11, (nmol, nl))
    atm.level.update(
        mixingratios={m: np.asarray(prof_f[m], dtype=np.float64) for m in sig.all_molecules}, temperature=tcol,
        pressure_bar=pcol, pressure=facet_form(f(0, (n,)), True), mmw=facet_form(f(1, (n,)), all_shared),
        den=facet_form(f(2, (n,)), False), z=facet_form(f(3, (n,)), False), dz=facet_form(f(4, (n,)), False),
        scale_height=facet_form(f(5, (n,)), False))
    atm.layer.update(
        mixingratios={m: facet_form(lmix[:, i], shared[i]) for i, m in enumerate(sig.all_molecules)},
        temperature=facet_form(f(6, (nl,)), False), pressure=facet_form(f(7, (nl,)), True),
        mmw=facet_form(f(8, (nl,)), all_shared), gravity=facet_form(f(9, (nl,)), False), colden=facet_form(f(10, (nl,)), False))
    c.nlevel, c.nlayer = n, nl
    atm.continuum_molecules = [list(p) for p in sig.continuum_molecules]
    atm.rayleigh_molecules = list(sig.rayleigh_molecules)
    atm.get_clouds(wno)
    if sig.no_opa:
        atm.add_warnings("I found chemistry for these but I do not have computed individual line "
                         "opacities (not including continuum) for: " + ",".join(sig.no_opa))
    atm.molecules = np.array(sig.molecules)
    ntot = nfac * nl
    rows = np.ascontiguousarray(ibuf[ioff[0]:ioff[1]].reshape(nfac, nopa, nl, 4).transpose(1, 0, 2, 3)).reshape(nopa, ntot, 4)
    wts = np.ascontiguousarray(f(12, (nopa, nl, 4)).transpose(1, 0, 2, 3)).reshape(nopa, ntot, 4)
    cia_rows = np.ascontiguousarray(ibuf[ioff[1]:ioff[2]].reshape(nfac, nc1, nl)[:, 0, :]).reshape(ntot)

    def tall(k, nsp):
        return np.ascontiguousarray(f(k, (nsp, nl)).transpose(1, 0, 2)).reshape(nsp, ntot)
    if sig.premixed:
        plan = dict(premixed=True, molecules=["premixed"], rows=rows, wts=wts, fac=np.ones(1), nlayer=ntot,
                    cia_pairs=list(sig.cia_pairs), cia_rows=rows2, cia_wts=wts2)
    else:
        plan = dict(molecules=list(sig.molecules), rows=rows, wts=wts, fac=np.ones(nopa), cia_pairs=list(sig.cia_pairs),
                    cia_rows=cia_rows, nlayer=ntot)
    factors = (tall(13, nopa), tall(14, ncont), list(sig.ray_names), tall(15, nray))
    atm._fast_tall = (plan, factors, opa)
    return atm
