"""Opacity retrieval + mixing on the GPU (counterpart of the reference ``picaso/optics.py``).

* ``RetrieveOpacities`` keeps the reference class's surface (``get_opacities(atm)``, ``wno``,
  ``nwno``, ``molecules``, ``rayleigh_molecules``, ``gauss_wts`` ...; reference optics.py:1877-2402)
  but loads the sqlite rows ONCE into HBM-resident tables (288 GB is room for a full
  1060-point x R~15000 monochromatic DB); ``get_opacities`` only bracket-searches the (P,T) grid on
  the host (a few hundred scalars) -- the per-(layer, wavelength) interpolation
  ``10**(sum w log10 kappa) * N_A`` runs in ``k_opacity_gas``.
* ``compute_opacity`` keeps the reference signature and 13-array return tuple (optics.py:26-431);
  the gas / Rayleigh sums and the mixing / delta-Eddington algebra run in ``k_opacity_gas`` and
  ``k_compute_opacity``.  ``compute_opacity_resident`` returns the planes as ``DeviceArray``s so that
  ``picaso()`` can hand them straight to the solvers without a PCIe round trip.

Database I/O (sqlite, ``np.load`` of blobs) and the Rayleigh cross sections are host-side data
preparation, out of the accelerated path: Rayleigh ``sigma(nu)`` per species is
supplied by the caller (``rayleigh_opa``) or read from an optional ``rayleigh`` table of the DB.
"""
import ctypes
import io
import math
import os
import sqlite3

import numpy as np

from . import _lib
from ._lib import check, f64, load, ptr
from .atmsetup import CloudTables
from .device import DeviceArray, regrid_rows
from .options import current as _options

_ci, _cd = ctypes.c_int, ctypes.c_double
AVOGADRO = 6.02214086e+23
OUT_NAMES = ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray", "gcos2", "dtau_og", "tau_og",
             "w0_og", "cosb_og", "w0_no_raman", "f_deltaM")


def _convert_array(text):
    out = io.BytesIO(text)
    out.seek(0)
    return np.load(out).copy()


def find_nearest(array, value):
    return int((np.abs(np.asarray(array) - value)).argmin())


class RetrieveOpacities:
    """Monochromatic opacity tables resident in HBM (reference ``RetrieveOpacities``,
    optics.py:1877-2402).  Build with ``from_sqlite`` (reference DB schema: tables ``header``,
    ``molecular``, ``continuum``) or ``from_arrays``."""

    def __init__(self, wno, pt_pairs, molecular, continuum, cia_temps, rayleigh_opa=None,
                 query_method="nearest", relative_flux=None, ctx=None):
        self.ctx = ctx if ctx is not None else _lib.context()
        self.ngauss = 1                                     # optics.py:1946-1947
        self.gauss_wts = np.array([1])
        self.wno = f64(wno)
        self.wave = 1e4 / self.wno
        self.nwno = int(self.wno.size)
        self.pt_pairs = sorted(set((int(i), float(p), float(t)) for i, p, t in pt_pairs),
                               key=lambda x: x[0])          # optics.py:2019
        temps = []
        for _, _, t in self.pt_pairs:
            if t not in temps:
                temps.append(t)
        self.temps = np.array(temps)
        self.nc_p = np.array([sum(1 for x in self.pt_pairs if x[2] == t) for t in temps])
        pres = []
        for _, p, _ in self.pt_pairs:
            if p not in pres:
                pres.append(p)
        self.pressures = np.array(pres)
        self.p_log_grid = np.log10(self.pressures)          # optics.py:2023-2025
        self.t_inv_grid = 1 / self.temps
        self.molecules = np.array(sorted(molecular.keys()))
        self.avail_continuum = sorted(continuum.keys())
        self.cia_temps = np.unique(np.asarray(cia_temps, dtype=float))
        if query_method not in ("nearest", "linear"):
            raise Exception("Do not recognize query method for opacities: %s. Options are nearest "
                            "or linear" % query_method)
        self.query_method = query_method
        self.relative_flux = relative_flux
        self.raman_stellar_shifts = None
        # ---- HBM-resident tables ----
        ptid = [x[0] for x in self.pt_pairs]
        self._row_of_ptid = {pid: r for r, pid in enumerate(ptid)}
        self._row_lut = np.full(int(max(ptid)) + 1, -1, dtype=np.int64)      # ptid -> table row
        self._row_lut[np.asarray(ptid, dtype=np.int64)] = np.arange(len(ptid))
        self._mol_raw, self._mol_log = {}, {}
        for m in self.molecules:
            tab = np.stack([f64(molecular[m][pid]) for pid in ptid])          # (npt, nwno)
            self._mol_raw[m] = DeviceArray.from_host(tab, self.ctx)
            if query_method == "linear":                    # optics.py:2281-2288 done once
                self._mol_log[m] = DeviceArray.from_host(
                    np.log10(np.where(tab != 0, tab, 1e-50)), self.ctx)
        self._cia = {}
        for pair in self.avail_continuum:
            tab = np.stack([f64(continuum[pair][t]) for t in self.cia_temps])
            self._cia[pair] = DeviceArray.from_host(tab, self.ctx)
        self.rayleigh_opa = {k: f64(v) for k, v in (rayleigh_opa or {}).items()}
        self.rayleigh_molecules = list(self.rayleigh_opa.keys())
        self._ray = {k: DeviceArray.from_host(v, self.ctx) for k, v in self.rayleigh_opa.items()}
        self.molecular_opa, self.continuum_opa = {}, {}
        self._plan = None

    # ------------------------------------------------------------------------------------------
    @classmethod
    def from_sqlite(cls, db_filename, wave_range=None, resample=1, query_method="nearest",
                    rayleigh_opa=None, ctx=None):
        """Read a monochromatic opacity DB in the reference schema (optics.py:1998-2039, 2159-2239)."""
        conn = sqlite3.connect(db_filename)
        cur = conn.cursor()
        cur.execute("SELECT wavenumber_grid FROM header")
        wno = _convert_array(cur.fetchone()[0])[::resample]
        wave = 1e4 / wno
        loc = slice(None) if wave_range is None else np.where(
            (wave > min(wave_range)) & (wave < max(wave_range)))
        cur.execute("SELECT ptid, pressure, temperature FROM molecular")
        pt_pairs = sorted(set(cur.fetchall()), key=lambda x: x[0])
        molecular = {}
        cur.execute("SELECT molecule, ptid, opacity FROM molecular")
        for mol, pid, blob in cur.fetchall():
            molecular.setdefault(mol, {})[int(pid)] = _convert_array(blob)[::resample][loc]
        continuum, cia_temps = {}, set()
        cur.execute("SELECT molecule, temperature, opacity FROM continuum")
        for mol, t, blob in cur.fetchall():
            continuum.setdefault(mol, {})[float(t)] = _convert_array(blob)[::resample][loc]
            cia_temps.add(float(t))
        if rayleigh_opa is None:
            try:
                cur.execute("SELECT molecule, opacity FROM rayleigh")
                rayleigh_opa = {m: _convert_array(b)[::resample][loc] for m, b in cur.fetchall()}
            except sqlite3.OperationalError:
                rayleigh_opa = {}
        conn.close()
        return cls(wno[loc], pt_pairs, molecular, continuum, sorted(cia_temps), rayleigh_opa,
                   query_method, ctx=ctx)

    # ------------------------------------------------------------------------------------------
    def find_needed_pts(self, tlayer, player):
        """Bracketing (1/T, log10 P) table rows and weights per layer; restates the index logic
        of reference optics.py:2048-2123 (ragged grid: ``nc_p`` pressures per temperature, row
        index = sum(nc_p[:it]) + ip, pressure index clamped to ``nc_p[it_hi] - 3``)."""
        t_inv = 1 / np.asarray(tlayer, dtype=float)
        p_log = np.log10(np.asarray(player, dtype=float))
        t_inv_grid, p_log_grid, nc_p = self.t_inv_grid, self.p_log_grid, self.nc_p
        # last grid index satisfying the condition (0 when none does), all layers at once
        def last_true(mask):
            return np.where(mask, np.arange(mask.shape[1])[None, :], -1).max(axis=1).clip(min=0)
        t_low_ind = last_true(t_inv_grid[None, :] > t_inv[:, None])
        t_low_ind[t_low_ind == (len(t_inv_grid) - 1)] = len(t_inv_grid) - 2
        t_hi_ind = t_low_ind + 1
        t_inv_low, t_inv_hi = t_inv_grid[t_low_ind], t_inv_grid[t_hi_ind]
        p_low_ind = last_true(p_log_grid[None, :] <= p_log[:, None])
        p_low_ind = np.minimum(p_low_ind, nc_p[t_hi_ind] - 3)
        p_log_low = p_log_grid[p_low_ind]
        p_hi_ind = p_low_ind + 1
        p_log_hi = p_log_grid[p_hi_ind]
        csum = np.concatenate([[0], np.cumsum(nc_p)])
        t_low_10XX, t_hi_10XX = csum[t_low_ind], csum[t_hi_ind]
        t_interp = ((t_inv - t_inv_low) / (t_inv_hi - t_inv_low))[:, np.newaxis]
        p_interp = ((p_log - p_log_low) / (p_log_hi - p_log_low))[:, np.newaxis]
        return (t_interp, p_interp, t_low_10XX + p_low_ind, t_hi_10XX + p_low_ind,
                t_low_10XX + p_hi_ind, t_hi_10XX + p_hi_ind)

    def get_opacities(self, atmosphere, exclude_mol=1):
        """Select table rows / weights for this atmosphere (reference optics.py:2241-2368).  The
        per-wavelength arithmetic is deferred to the GPU (``compute_opacity``)."""
        fast = getattr(atmosphere, "_fast", None)
        if fast is not None and fast[2] is self and exclude_mol == 1:       # formed with the atmosphere (fastsetup.py)
            self._plan = fast[0]
            self.molecular_opa = _LazyPlanes(self, "mol")
            self.continuum_opa = _LazyPlanes(self, "cia")
            return
        nlayer = atmosphere.c.nlayer
        tlayer = np.asarray(atmosphere.layer["temperature"], dtype=float)
        player = np.asarray(atmosphere.layer["pressure"], dtype=float) / atmosphere.c.pconv
        molecules = [m for m in atmosphere.molecules]
        cia_pairs = [k[0] + k[1] for k in atmosphere.continuum_molecules]
        rows = np.zeros((len(molecules), nlayer, 4), dtype=np.int32)
        wts = np.zeros((len(molecules), nlayer, 4))
        if self.query_method == "linear":
            t_i, p_i, i_ll, i_hl, i_lh, i_hh = self.find_needed_pts(tlayer, player)
            t_i, p_i = t_i[:, 0], p_i[:, 0]
            # order of the reference's four terms (optics.py:2290-2293)
            r4 = np.stack([i_ll, i_hl, i_hh, i_lh], axis=1)
            w4 = np.stack([(1 - t_i) * (1 - p_i), t_i * (1 - p_i), t_i * p_i, (1 - t_i) * p_i], axis=1)
            # the reference addresses rows by ptid = 1 + index; tables are stored in ptid order
            ids = 1 + r4
            lut = self._row_lut[np.minimum(ids, len(self._row_lut) - 1)]
            missing = (ids >= len(self._row_lut)) | (lut < 0)
            if missing.any():
                raise KeyError("opacity table has no row for ptid %d" % int(ids[missing][0]))
            rows[:] = lut[None]
            wts[:] = w4[None]
            atmosphere.layer["pt_opa_index"] = 1 + np.unique(r4)
        else:                                               # optics.py:2330-2332
            ind_pt = [min(self.pt_pairs, key=lambda c: math.hypot(
                np.log(c[1]) - np.log(coordinate[0]), c[2] - coordinate[1]))[0]
                for coordinate in zip(player, tlayer)]
            rows[:, :, 0] = np.array([self._row_of_ptid[i] for i in ind_pt])[None]
            wts[:, :, 0] = 1.0
            atmosphere.layer["pt_opa_index"] = ind_pt
        fac = np.ones(len(molecules))
        if exclude_mol != 1:
            fac = np.array([exclude_mol[m] for m in molecules], dtype=float)
        temps = np.unique(self.cia_temps)
        cia_rows = np.abs(temps[None, :] - tlayer[:, None]).argmin(axis=1).astype(np.int32)   # :2298
        self._plan = dict(molecules=molecules, rows=rows, wts=wts, fac=fac, cia_pairs=cia_pairs,
                          cia_rows=cia_rows, nlayer=nlayer)
        self.molecular_opa = _LazyPlanes(self, "mol")
        self.continuum_opa = _LazyPlanes(self, "cia")

    get_opacities_nearest = get_opacities

    # materialise one (nlayer, nwno) opacity plane on request (debugging / drop-in access)
    def _materialise(self, kind, key):
        pl = self._plan
        nlayer = pl["nlayer"]
        one = np.ones((1, nlayer))
        zero_g = DeviceArray((nlayer, self.nwno), self.ctx)
        zero_r = DeviceArray((nlayer, self.nwno), self.ctx)
        if kind == "mol":
            m = pl["molecules"].index(key)
            fac = one * pl["fac"][m] / AVOGADRO             # kernel multiplies by N_A * fac
            tabs = [self._mol_log[key] if self.query_method == "linear" else self._mol_raw[key]]
            _gas_call(self, nlayer, tabs, pl["rows"][m:m + 1], pl["wts"][m:m + 1], fac * AVOGADRO,
                      [], None, None, [], None, zero_g, zero_r)
        else:
            _gas_call(self, nlayer, [], None, None, None, [self._cia[key]],
                      pl["cia_rows"][None], one, [], None, zero_g, zero_r)
        return zero_g.to_host()


class RetrieveCKs:
    """Pre-mixed correlated-k tables resident in HBM (reference ``RetrieveCKs`` with
    ``method='preweighted'``, optics.py:655-1547): ``ln kappa[p, t, wno, gauss]`` on a ragged
    (P,T) grid (``nc_p`` pressures per temperature) plus the CIA continuum.  ``get_opacities``
    restates the host index search of ``get_pre_mix_ck`` (:1081-1150) and ``get_continuum``
    (:1411-1428, 1474-1478); the exponentials of the interpolated logs run in ``k_opacity_gas``.

    Built from arrays (the reference's legacy ascii / HDF5 table readers are file-format code
    outside the accelerated path); ``continuum`` maps 'H2H2'-style keys to {temperature: kappa(wno)}.
    """

    def __init__(self, wno, gauss_wts, pressures, temps, nc_p, ln_kappa=None, continuum=None, cia_temps=(),
                 rayleigh_opa=None, relative_flux=None, ctx=None, kappas=None, gauss_pts=None, on_fly=None):
        self.ctx = ctx if ctx is not None else _lib.context()
        self.wno = f64(wno)
        self.wave = 1e4 / self.wno
        self.nwno = int(self.wno.size)
        self.gauss_wts = f64(gauss_wts)
        self.ngauss = int(self.gauss_wts.size)
        self.pressures = np.unique(np.asarray(pressures, dtype=float))     # optics.py:1095
        self.temps = np.unique(np.asarray(temps, dtype=float))             # optics.py:1097
        self.nc_p = np.asarray(nc_p, dtype=int)
        shape = (self.pressures.size, self.temps.size, self.nwno, self.ngauss)
        if ln_kappa is None and not kappas:
            raise Exception("RetrieveCKs: give the premixed ln_kappa table and/or per-gas kappas")
        self._kappa = None
        if ln_kappa is not None:
            k = f64(ln_kappa)
            if k.shape != shape:
                raise Exception("RetrieveCKs: ln_kappa must be (npres, ntemp, nwno, ngauss) = %s, got %s"
                                % (shape, k.shape))
            self._kappa = DeviceArray.from_host(k.reshape((-1, self.nwno * self.ngauss)), self.ctx)
        # per-gas ln(kappa) tables for on-the-fly mixing (the reference's self.kappas, optics.py:1315)
        self._kappas = {}
        for m, tab in (kappas or {}).items():
            t = f64(tab)
            if t.shape != shape:
                raise Exception("RetrieveCKs: kappas[%r] must be %s, got %s" % (m, shape, t.shape))
            self._kappas[m] = DeviceArray.from_host(t, self.ctx)
        self.gauss_pts = f64(gauss_pts) if gauss_pts is not None else None
        if self._kappas and (self.gauss_pts is None or self.gauss_pts.size != self.ngauss):
            raise Exception("RetrieveCKs: on-the-fly mixing needs gauss_pts (ngauss abscissae)")
        continuum = continuum or {}
        self.avail_continuum = sorted(continuum.keys())
        self.cia_temps = np.sort(np.unique(np.asarray(cia_temps, dtype=float)))
        self._cia = {}
        for pair in self.avail_continuum:                   # ln(kappa) rows in sorted-temperature order
            tab = np.stack([f64(continuum[pair][t]) for t in self.cia_temps])
            self._cia[pair] = DeviceArray.from_host(np.log(tab), self.ctx)
        self.rayleigh_opa = {m: f64(v) for m, v in (rayleigh_opa or {}).items()}
        self.rayleigh_molecules = list(self.rayleigh_opa.keys())
        self._ray = {m: DeviceArray.from_host(v, self.ctx) for m, v in self.rayleigh_opa.items()}
        self.molecules = np.array(list(self._kappas.keys()))  # premixed only: no per-molecule tables
        self.query_method = "premixed"
        self.relative_flux = relative_flux
        self.raman_stellar_shifts = None
        self.molecular_opa, self.continuum_opa = None, {}
        self._plan = None
        # the reference binds get_opacities to the on-the-fly or the preweighted variant at
        # construction (optics.py:688, :713); default: on the fly when only per-gas tables are given
        self.on_fly = bool(on_fly) if on_fly is not None else (self._kappa is None)
        if self.on_fly:
            if not self._kappas:
                raise Exception("RetrieveCKs: on_fly=True needs per-gas kappas")
            self.get_opacities = self.get_opacities_deq_onfly

    @classmethod
    def from_files(cls, ck_db, continuum_db, method="preweighted", preload_gases="all", rayleigh_opa=None,
                   ctx=None, refdata=None):
        """The reference's constructor ``RetrieveCKs(ck_dir, continuum_db, method, preload_gases)``
        (optics.py:676-722): ``method='preweighted'`` reads ONE premixed HDF5 table, ``'resortrebin'`` a directory
        of per-gas tables for on-the-fly mixing (``read_ck_tables``); the continuum comes from the sqlite database
        the reference ships (``read_continuum_db``).  The legacy ``ascii_data`` / ``full_abunds`` directory form
        (optics.py:772-1058; deprecated there) is not read.  Rayleigh cross sections are data supplied by the
        caller, as for the monochromatic tables."""
        if method not in ("preweighted", "resortrebin"):
            raise Exception("Only resortrebin and preweighted are options for Correlated-Ks")
        if method == "preweighted" and os.path.isdir(ck_db):
            raise Exception("method='preweighted' reads the premixed HDF5 file; the legacy ascii_data / full_abunds "
                            "directory of the reference (deprecated there) is not supported: %s" % ck_db)
        t = read_ck_tables(ck_db, preload_gases=preload_gases if method == "resortrebin" else None, refdata=refdata)
        cwno, continuum, cia_temps = read_continuum_db(continuum_db)
        if cwno.shape != np.shape(t["wno"]) or not np.allclose(cwno, t["wno"], rtol=1e-6):
            raise Exception("the continuum database %s is on a different wavenumber grid (%d points) than the "
                            "k-tables (%d)" % (continuum_db, cwno.size, np.size(t["wno"])))
        # ragged grid: pressure / temperature of every table point, temperature-major (optics.py:1095-1141)
        temps, nc_p = np.asarray(t["temps"], dtype=float), np.asarray(t["nc_p"], dtype=int)
        press = np.asarray(t["pressures"], dtype=float)
        pressures = np.concatenate([press[:n] for n in nc_p])
        temps_flat = np.concatenate([[tt] * n for tt, n in zip(np.unique(temps), nc_p)])
        kw = dict(continuum=continuum, cia_temps=cia_temps, rayleigh_opa=rayleigh_opa, ctx=ctx)
        if method == "preweighted":
            if "kappa" not in t:
                raise Exception("method='preweighted' needs the premixed HDF5 file, got a directory: %s" % ck_db)
            obj = cls(t["wno"], t["gauss_wts"], pressures, temps_flat, nc_p, ln_kappa=t["kappa"],
                      gauss_pts=t["gauss_pts"], **kw)
            obj.full_abunds = {k: t["abunds"][:, i] for i, k in enumerate(t["abunds_map"])}
        else:
            obj = cls(t["wno"], t["gauss_wts"], pressures, temps_flat, nc_p, kappas=t["kappas"],
                      gauss_pts=t["gauss_pts"], on_fly=True, **kw)
        obj.delta_wno = np.asarray(t["delta_wno"], dtype=float)
        obj.ck_filename, obj.continuum_db, obj.preload_gases = ck_db, continuum_db, t["molecules"]
        return obj

    def get_opacities(self, atmosphere, exclude_mol=1):
        """Table rows / weights for this atmosphere (``get_opacities_preweighted``: continuum +
        ``get_pre_mix_ck``, reference optics.py:1500-1538)."""
        if exclude_mol != 1:
            raise Exception("premixed correlated-k tables cannot exclude molecules")
        fast = getattr(atmosphere, "_fast", None)
        if fast is not None and fast[2] is self and fast[0].get("premixed"):    # formed with the atmosphere (fastsetup.py)
            self._plan = fast[0]
            self.continuum_opa = _LazyPlanes(self, "cia")
            self.molecular_opa = None
            return
        nlayer = atmosphere.c.nlayer
        if self._kappa is None:
            raise Exception("no premixed table loaded: use get_opacities_deq_onfly")
        (p_low, p_hi, t_low, t_hi), t_i, p_i = self.get_mixing_indices(atmosphere)
        nt = self.temps.size
        # the reference's four terms in order (optics.py:1153-1156); row = ip * ntemp + it
        rows = np.stack([p_low * nt + t_low, p_low * nt + t_hi, p_hi * nt + t_hi, p_hi * nt + t_low],
                        axis=1).astype(np.int32)[None]
        wts = np.stack([(1 - t_i) * (1 - p_i), t_i * (1 - p_i), t_i * p_i, (1 - t_i) * p_i], axis=1)[None]
        self._plan = dict(premixed=True, molecules=["premixed"], rows=rows, wts=wts, fac=np.ones(1), nlayer=nlayer)
        self._plan_continuum(atmosphere)
        self.molecular_opa = None

    def get_mixing_indices(self, atmosphere):
        """Bracketing pressure / temperature grid indices and the 1/T, log10 P interpolation weights
        of every layer (reference ``get_mixing_indices``, optics.py:1200-1278; the same search opens
        ``get_pre_mix_ck``, :1081-1150).  Returns ``[p_low, p_hi, t_low, t_hi], t_interp, p_interp``."""
        nlayer = atmosphere.c.nlayer
        tlayer = np.asarray(atmosphere.layer["temperature"], dtype=float)
        player = np.asarray(atmosphere.layer["pressure"], dtype=float) / atmosphere.c.pconv
        t_inv, p_log = 1 / tlayer, np.log10(player)
        p_log_grid = np.log10(self.pressures[self.pressures > 0])
        t_inv_grid = 1 / self.temps

        # last grid index satisfying the condition, 0 when none does (optics.py:1101-1113, 1124-1141).  Both grids are
        # strictly monotonic (np.unique in __init__), so "the last index with grid > x" / "<= x" is a count: a binary
        # search instead of an (nlayer, ngrid) mask -- nlayer is nfacets * nlayer for the tall atmosphere of a 3-D spectrum
        asc = t_inv_grid[::-1]                                                   # 1/T of ascending temperatures: descending
        t_low = (len(asc) - np.searchsorted(asc, t_inv, side="right") - 1).clip(min=0)      # last index with grid > 1/T
        t_low[t_low == (len(t_inv_grid) - 1)] = len(t_inv_grid) - 2
        t_hi = t_low + 1
        p_low = (np.searchsorted(p_log_grid, p_log, side="right") - 1).clip(min=0)          # last index with grid <= log10 P
        p_low = np.minimum(p_low, self.nc_p[t_hi] - 3)
        p_hi = p_low + 1
        t_i = (t_inv - t_inv_grid[t_low]) / (t_inv_grid[t_hi] - t_inv_grid[t_low])
        p_i = (p_log - p_log_grid[p_low]) / (p_log_grid[p_hi] - p_log_grid[p_low])
        return np.array([p_low, p_hi, t_low, t_hi]), t_i, p_i

    def mix_my_opacities_gasesfly(self, atmosphere, exclude_mol=1):
        """On-the-fly mixing of the per-gas k-tables at the four P-T neighbours of every layer
        (reference optics.py:1164-1198): ``k_ckmix`` leaves ln of the mixed coefficients in HBM as
        ``4*nlayer`` table rows; the ln-bilinear interpolation and exp (:1191-1197) then run in
        ``k_opacity_gas`` exactly as for a premixed table."""
        from . import resident
        nlayer = atmosphere.c.nlayer
        mols = [m for m in atmosphere.molecules if (exclude_mol == 1) or (exclude_mol[m] == 1)]
        if not mols:
            raise Exception("mix_my_opacities_gasesfly: no molecules to mix")
        for m in mols:
            if m not in self._kappas:
                raise KeyError("no k-table for %s" % m)
        mix = atmosphere.layer["mixingratios"]
        mixes = [np.asarray(mix[m].values if hasattr(mix[m], "values") else mix[m], dtype=float) for m in mols]
        indices, t_i, p_i = self.get_mixing_indices(atmosphere)
        self._mixed = resident.mix_all_gases_gasesfly(self.ctx, [self._kappas[m] for m in mols], mixes,
                                                      self.gauss_pts, self.gauss_wts, indices)
        base = 4 * np.arange(nlayer)
        # (1-t)(1-p) k[0] + t(1-p) k[1] + t p k[3] + (1-t) p k[2]       (optics.py:1193-1196)
        rows = np.stack([base, base + 1, base + 3, base + 2], axis=1).astype(np.int32)[None]
        wts = np.stack([(1 - t_i) * (1 - p_i), t_i * (1 - p_i), t_i * p_i, (1 - t_i) * p_i], axis=1)[None]
        keep = {k: self._plan[k] for k in ("cia_pairs", "cia_rows", "cia_wts")} if self._plan else {}
        self._plan = dict(premixed=True, molecules=["premixed"], rows=rows, wts=wts, fac=np.ones(1), nlayer=nlayer,
                          table=self._mixed.reshape((4 * nlayer, self.nwno * self.ngauss)), **keep)
        self.molecular_opa = None

    def get_opacities_deq_onfly(self, atmosphere, exclude_mol=1):
        """Continuum + on-the-fly mixed molecular opacity (reference optics.py:1512-1525)."""
        self.mix_my_opacities_gasesfly(atmosphere, exclude_mol=exclude_mol)
        self._plan_continuum(atmosphere)

    def _plan_continuum(self, atmosphere):
        # continuum: bracketing CIA temperatures and the 1/T weight (optics.py:1411-1428, 1474-1478)
        nlayer = atmosphere.c.nlayer
        tlayer = np.asarray(atmosphere.layer["temperature"], dtype=float)
        st = self.cia_temps
        cia_pairs = [k[0] + k[1] for k in atmosphere.continuum_molecules]
        cia_rows = np.zeros((nlayer, 2), dtype=np.int32)
        cia_wts = np.zeros((nlayer, 2))
        if cia_pairs:
            lo = np.searchsorted(st, tlayer, side="right") - 1      # last CIA temperature <= T_layer (st is sorted, unique)
            lo = np.clip(lo, 0, len(st) - 2)              # below the grid: first pair; at/above its top: last pair
            ti = (1 / tlayer - 1 / st[lo]) / (1 / st[lo + 1] - 1 / st[lo])
            cia_rows = np.stack([lo, lo + 1], axis=1).astype(np.int32)
            cia_wts = np.stack([1 - ti, ti], axis=1)
        self._plan.update(cia_pairs=cia_pairs, cia_rows=cia_rows, cia_wts=cia_wts)
        self.continuum_opa = _LazyPlanes(self, "cia")

    def get_opacities_preweighted(self, atmosphere, exclude_mol=1):
        return RetrieveCKs.get_opacities(self, atmosphere, exclude_mol)

    def get_molecular_opa(self):
        """``molecular_opa`` (nlayer, nwno, ngauss) as the reference stores it (optics.py:1159)."""
        pl = self._plan
        nlayer = pl["nlayer"]
        tg = DeviceArray((nlayer, self.nwno, self.ngauss), self.ctx)
        tr = DeviceArray((nlayer, self.nwno), self.ctx)
        _gas_call(self, nlayer, [pl.get("table", self._kappa)], pl["rows"], pl["wts"], np.ones((1, nlayer)), [], None,
                  None, [], None, tg, tr, mol_mode=2, ngauss=self.ngauss)
        return tg.to_host()

    def _materialise(self, kind, key):
        pl = self._plan
        nlayer = pl["nlayer"]
        tg = DeviceArray((nlayer, self.nwno), self.ctx)
        tr = DeviceArray((nlayer, self.nwno), self.ctx)
        _gas_call(self, nlayer, [], None, None, None, [self._cia[key]], pl["cia_rows"][None],
                  np.ones((1, nlayer)), [], None, tg, tr, mol_mode=0, cont_wts=pl["cia_wts"][None], ngauss=1)
        return tg.to_host()


# ------------------------------------------------------------------------------------------------
# correlated-k table files -> arrays (the readers in front of the device-resident tables)
# ------------------------------------------------------------------------------------------------
def g_w_2gauss(order=4, gfrac=0.95):
    """Abscissae and weights of the two-part Gauss quadrature of the k-tables (reference
    opacity_factory.py:1474-1503): ``order`` Gauss-Legendre points on [0, gfrac] and on [gfrac, 1]."""
    g, w = np.polynomial.legendre.leggauss(order)
    return (np.concatenate((gfrac * 0.5 * (g + 1.0), gfrac + (1.0 - gfrac) * 0.5 * (g + 1.0))),
            np.concatenate((gfrac * w * 0.5, (1.0 - gfrac) * w * 0.5)))


def _h5py():
    try:
        import h5py
    except ImportError as e:       # not a dependency of the solver path: only these readers need it
        raise Exception("reading HDF5 correlated-k tables needs the h5py package (pip install h5py); the "
                        "tables can also be handed to RetrieveCKs as arrays") from e
    return h5py


def read_continuum_db(continuum_db):
    """``(wno, {pair: {T: kappa(wno)}}, cia_temps)`` from the sqlite continuum database the reference ships
    for its k-tables (tables ``header`` / ``continuum``; reference optics.py:1067-1079, 1398-1498)."""
    if not os.path.isfile(continuum_db):
        raise Exception("The continuum opacity file does not exist: %s" % continuum_db)
    conn = sqlite3.connect(continuum_db)
    cur = conn.cursor()
    cur.execute("SELECT wavenumber_grid FROM header")
    wno = _convert_array(cur.fetchone()[0])
    continuum, cia_temps = {}, set()
    cur.execute("SELECT molecule, temperature, opacity FROM continuum")
    for mol, t, blob in cur.fetchall():
        continuum.setdefault(mol, {})[float(t)] = _convert_array(blob)
        cia_temps.add(float(t))
    conn.close()
    return wno, continuum, np.array(sorted(cia_temps))


def read_ck_tables(path, preload_gases=None, refdata=None):
    """The reference's ``get_ck_tables`` (opacity_factory.py:2221-2327, the successor of
    ``RetrieveCKs.get_h5_data``, optics.py:725-770, and ``load_kcoeff_arrays_first``): either ONE premixed HDF5
    file (datasets ``ck_molecules, wno, delta_wno, pressures, temperatures, gauss_pts, gauss_wts, kcoeffs
    [npres, ntemp, nwno, ngauss] = ln kappa, abunds, abunds_map``; ``nc_p`` = table points per temperature) or
    a DIRECTORY of per-gas tables for on-the-fly mixing (``<gas>_1460.hdf5`` with the same datasets plus ``nc_p``,
    or ``<gas>_1460.npy`` arrays on the 661-bin grid, whose axes come from ``$picaso_refdata``:
    ``climate_INPUTS/wvno_661`` and ``opacities/grid1460.csv``, with the 2 x 4-point Gauss set).  Returns plain
    numpy arrays: ``wno, delta_wno, pressures, temps, nc_p, gauss_pts, gauss_wts, molecules`` and ``kappa``
    (premixed) or ``kappas`` {gas: table}."""
    out = {}
    if os.path.isfile(path) and ((".hdf5" in path) or (".h5" in path)):
        h5py = _h5py()
        with h5py.File(path, "r") as f:
            out["molecules"] = [x.decode("utf-8") for x in f["ck_molecules"][:]]
            out["wno"], out["delta_wno"] = f["wno"][:], f["delta_wno"][:]
            pres, temp = np.asarray(f["pressures"][:], dtype=float), np.asarray(f["temperatures"][:], dtype=float)
            out["gauss_pts"], out["gauss_wts"] = f["gauss_pts"][:], f["gauss_wts"][:]
            out["kappa"] = f["kcoeffs"][:]
            out["abunds"] = np.asarray(f["abunds"][:])
            out["abunds_map"] = [x.decode("utf-8") for x in f["abunds_map"][:]]
        # table points per temperature, temperatures in increasing order (pandas groupby('temperature').size())
        ut = np.unique(temp)
        out["nc_p"] = np.array([int(np.sum(temp == t)) for t in ut])
        out["temps"], out["pressures"] = ut, np.unique(pres)
        return out
    if not os.path.isdir(path):
        raise Exception("The CK filename that you have selected does not exist. Please make sure you have "
                        "downloaded and unpacked the right CK file.")
    import glob
    if preload_gases is None or (isinstance(preload_gases, str) and preload_gases == "all"):
        found = sorted(glob.glob(os.path.join(path, "*.hdf5"))) or sorted(glob.glob(os.path.join(path, "*.npy")))
        if not found:
            raise Exception("No .npy or .hdf5 molecule files were found in %s" % path)
        preload_gases = [os.path.basename(f).split("_")[0] for f in found]       # justdoit.py:1397-1407
    elif isinstance(preload_gases, str):
        preload_gases = [preload_gases]
    out["kappas"] = {}
    for mol in preload_gases:
        f5, fn = os.path.join(path, "%s_1460.hdf5" % mol), os.path.join(path, "%s_1460.npy" % mol)
        if os.path.isfile(f5):
            h5py = _h5py()
            with h5py.File(f5, "r") as f:
                out["wno"], out["delta_wno"] = f["wno"][:], f["delta_wno"][:]
                out["pressures"], out["temps"] = np.unique(f["pressures"][:]), np.unique(f["temperatures"][:])
                out["gauss_pts"], out["gauss_wts"] = f["gauss_pts"][:], f["gauss_wts"][:]
                out["nc_p"] = np.array([int(i) for i in f["nc_p"][:]])
                out["kappas"][mol] = f["kcoeffs"][:]
        elif os.path.isfile(fn):
            ref = refdata or os.environ.get("picaso_refdata")
            if ref is None:
                raise Exception("the .npy k-tables take their axes from $picaso_refdata (climate_INPUTS/wvno_661, "
                                "opacities/grid1460.csv): set the picaso_refdata environment variable")
            out["wno"], out["delta_wno"] = np.loadtxt(os.path.join(ref, "climate_INPUTS", "wvno_661"),
                                                      usecols=[0, 1], unpack=True)
            grid = np.genfromtxt(os.path.join(ref, "opacities", "grid1460.csv"), delimiter=",", names=True)
            t_all, p_all = np.asarray(grid["temperature_K"], dtype=float), np.asarray(grid["pressure_bar"], dtype=float)
            # pandas .unique(): order of first appearance; groupby().size(): increasing temperature
            _, ip = np.unique(p_all, return_index=True)
            _, it = np.unique(t_all, return_index=True)
            out["pressures"], out["temps"] = p_all[np.sort(ip)], t_all[np.sort(it)]
            out["nc_p"] = np.array([int(np.sum(t_all == t)) for t in np.unique(t_all)])
            out["gauss_pts"], out["gauss_wts"] = g_w_2gauss(order=4, gfrac=0.95)
            out["kappas"][mol] = np.load(fn)
        else:                                       # the reference says so and goes on (opacity_factory.py:2286-2290)
            import warnings
            warnings.warn("no k-table for %s in %s (neither %s_1460.hdf5 nor %s_1460.npy): it is left out of the "
                          "mixing" % (mol, path, mol, mol), UserWarning)
    if not out["kappas"]:
        raise Exception("Uh oh. No molecules are left to mix. Its likely you have not downloaded the correct files.")
    out["molecules"] = list(out["kappas"].keys())
    return out


def shard_opacity(opa, lo, hi, ctx):
    """The wavelength block ``[lo, hi)`` of an opacity object (``RetrieveOpacities`` or ``RetrieveCKs``) with
    its tables resident on ANOTHER context / GPU: the multi-GPU form of ``picaso()`` (``devices=N``) cuts the
    grid of one spectrum into contiguous blocks and keeps each block's tables on its own device.
    Every function on the path is pointwise in wavelength, so a shard is a complete opacity object of ``hi - lo``
    wavelengths; per-wavelength host vectors are sliced, tables are sliced along their wavelength axis (Gauss
    index fastest inside a wavelength for correlated-k tables) and uploaded once.  ``[0, nwno)`` on the object's
    own context is the object itself."""
    import copy
    lo, hi = int(lo), int(hi)
    if not (0 <= lo < hi <= opa.nwno):
        raise Exception("shard_opacity: block [%d, %d) outside the %d-point grid" % (lo, hi, opa.nwno))
    same_ctx = getattr(ctx, "value", ctx) == getattr(opa.ctx, "value", opa.ctx)
    if lo == 0 and hi == opa.nwno and same_ctx:
        return opa
    nwno, ng = opa.nwno, opa.ngauss
    s = copy.copy(opa)
    s.ctx = ctx
    s.nwno = hi - lo
    s.wno = np.ascontiguousarray(opa.wno[lo:hi])
    s.wave = 1e4 / s.wno
    s._plan = None
    s.__dict__.pop("_resident_vectors", None)
    s.__dict__.pop("_shards", None)
    s.__dict__.pop("_replicas", None)
    s.__dict__.pop("_const_planes", None)
    s.__dict__.pop("_d_wno", None)
    s.__dict__.pop("_raman_pollack", None)
    s.__dict__.pop("_raman_oklopcic", None)
    s.__dict__.pop("_trapz", None)
    s.__dict__.pop("_trapz_buf", None)
    s.__dict__.pop("_driver_tables", None)
    s.__dict__.pop("_cloud_tables_dev", None)
    for k in ("_trapz_dev", "_bond_denom", "_bond_denom_host", "_ones", "_fast_setup"):
        s.__dict__.pop(k, None)
    s.molecular_opa, s.continuum_opa = ({} if isinstance(opa.molecular_opa, dict) else None), {}

    def cols(d, per=1):
        """device table (..., nwno*per) -> its columns of the block, on ctx"""
        if d is None:
            return None
        shape = tuple(d.shape)
        h = d.to_host().reshape((-1, nwno * per))
        out = DeviceArray.from_host(np.ascontiguousarray(h[:, lo * per:hi * per]), ctx)
        if len(shape) == 1:
            return out.reshape(((hi - lo) * per,))
        return out
    for name in ("_mol_raw", "_mol_log", "_cia"):
        if hasattr(opa, name):
            setattr(s, name, {k: cols(v) for k, v in getattr(opa, name).items()})
    if getattr(opa, "_kappa", None) is not None:
        s._kappa = cols(opa._kappa, ng)
    if hasattr(opa, "_kappas"):
        s._kappas = {}
        for m, tab in opa._kappas.items():
            npres, ntemp = tab.shape[0], tab.shape[1]
            s._kappas[m] = cols(tab, ng).reshape((npres, ntemp, hi - lo, ng))
    s.rayleigh_opa = {k: np.ascontiguousarray(v[lo:hi]) for k, v in opa.rayleigh_opa.items()}
    s._ray = {k: DeviceArray.from_host(v, ctx) for k, v in s.rayleigh_opa.items()}
    s._parent_stamp = None
    resync_shard(s, opa, lo, hi)
    if hasattr(opa, "get_opacities") and getattr(opa, "on_fly", False):
        s.get_opacities = s.get_opacities_deq_onfly          # bound method of the shard, not of the parent
    return s


# host attributes callers (re)assign on the opacity object AFTER a shard of it may exist: star() hangs the stellar
# spectrum and the Raman shift ratios there, opannection's docstring lets query_method / raman_db be set later
_SHARD_VECTORS = ("relative_flux", "unshifted_stellar_spec", "delta_wno", "raman_stellar_shifts")
_SHARD_SCALARS = ("raman_db", "query_method")


def resync_shard(s, opa, lo, hi):
    """Bring a shard's copies of the parent's mutable host attributes up to date (called on every use of a cached
    shard or replica: ``picaso(devices=...)``, ``phase_curve(devices=...)``).  The parent's arrays are compared by a digest
    of their whole content (``content_digest``; 0.1 ms for the 0.8 MB vectors of a 1e5-point grid), so ``star()`` called
    again, an array replaced or one edited in place all reach the shard."""
    if s is opa:
        return s
    def mark(v):          # arrays by content (every byte: an in-place edit of the stellar spectrum counts), the rest by value
        if isinstance(v, np.ndarray):
            return content_digest(v)
        return v if isinstance(v, (str, int, float, type(None))) else id(v)
    stamp = tuple(mark(getattr(opa, n, None)) for n in _SHARD_VECTORS + _SHARD_SCALARS)
    if getattr(s, "_parent_stamp", None) == stamp:
        return s
    nwno = opa.nwno
    for name in _SHARD_VECTORS:
        v = getattr(opa, name, None)
        if isinstance(v, np.ndarray) and v.shape[:1] == (nwno,):
            setattr(s, name, np.ascontiguousarray(v[int(lo):int(hi)]))
        elif hasattr(opa, name):
            setattr(s, name, v)
        else:
            s.__dict__.pop(name, None)
    for name in _SHARD_SCALARS:
        if hasattr(opa, name):
            setattr(s, name, getattr(opa, name))
    # device tables derived from them on the shard
    s.__dict__.pop("_raman_oklopcic", None)
    s.__dict__.pop("_resident_vectors", None)
    s._plan = None                         # the (P, T) plan is rebuilt by the next get_opacities call
    s._parent_stamp = stamp
    return s


class _LazyPlanes(dict):
    """``molecular_opa`` / ``continuum_opa`` dictionaries whose planes are computed on first access."""

    def __init__(self, opa, kind):
        super().__init__()
        self._opa, self._kind = opa, kind
        names = opa._plan["molecules"] if kind == "mol" else opa._plan["cia_pairs"]
        self._names = list(names)

    def keys(self):
        return self._names

    def __iter__(self):
        return iter(self._names)

    def __len__(self):
        return len(self._names)

    def __contains__(self, k):
        return k in self._names

    def __getitem__(self, k):
        if not dict.__contains__(self, k):
            if k not in self._names:
                raise KeyError(k)
            dict.__setitem__(self, k, self._opa._materialise(self._kind, k))
        return dict.__getitem__(self, k)


def _ptr_array(devs):
    arr = (ctypes.c_void_p * max(1, len(devs)))()
    for i, d in enumerate(devs):
        arr[i] = d.addr
    return ctypes.cast(arr, ctypes.POINTER(ctypes.POINTER(ctypes.c_double)))


def _gas_call(opa, nlayer, mol_tabs, mol_rows, mol_wts, mol_fac, cont_tabs, cont_rows, cont_fac,
              ray_tabs, ray_fac, taugas, tauray, mol_mode=None, cont_wts=None, ngauss=1, mix=None):
    """``picaso_opacity_gas_ck_dev``: gather + interpolate table rows into TAUGAS / TAURAY.
    ``mol_mode`` 0 nearest, 1 log10-bilinear (monochromatic 'linear'), 2 ln-bilinear (premixed CK);
    ``cont_wts`` given -> log-linear continuum between two rows per layer (CK)."""
    ip = ctypes.POINTER(ctypes.c_int)

    def iarr(x):
        return None if x is None else np.ascontiguousarray(x, dtype=np.int32)

    if mol_mode is None:
        mol_mode = 1 if opa.query_method == "linear" else 0
    mr, cr = iarr(mol_rows), iarr(cont_rows)
    mw = f64(mol_wts) if mol_wts is not None else None
    mf = f64(mol_fac) if mol_fac is not None else None
    cw = f64(cont_wts) if cont_wts is not None else None
    cf = f64(cont_fac) if cont_fac is not None else None
    rf = f64(ray_fac) if ray_fac is not None else None
    gas_args = (_ci(mol_mode),
                _ci(len(mol_tabs)), _ptr_array(mol_tabs), mr.ctypes.data_as(ip) if mr is not None else None,
                ptr(mw), ptr(mf), _ci(1 if cw is not None else 0), _ci(len(cont_tabs)), _ptr_array(cont_tabs),
                cr.ctypes.data_as(ip) if cr is not None else None, ptr(cw), ptr(cf), _ci(len(ray_tabs)),
                _ptr_array(ray_tabs), ptr(rf))
    if mix is not None:       # gas stage + compute_opacity in one launch (ngauss = 1): `mix` = the mixing's arguments
        check(load().picaso_gas_compute_opacity_dev(opa.ctx, _ci(nlayer), _ci(opa.nwno), *gas_args, *mix), opa.ctx)
        return
    check(load().picaso_opacity_gas_ck_dev(
        opa.ctx, _ci(nlayer), _ci(opa.nwno), _ci(ngauss), *gas_args, ptr(taugas.addr), ptr(tauray.addr)), opa.ctx)


def _layer_factors(atm, opacityclass):
    """Per-layer scalar coefficients of the TAUGAS / TAURAY sums (reference optics.py:144-277)."""
    fast = getattr(atm, "_fast", None)
    if fast is not None and fast[2] is opacityclass and opacityclass._plan is fast[0]:
        return fast[1]
    pl = opacityclass._plan
    nlayer = atm.c.nlayer
    tlevel = np.asarray(atm.level["temperature"], dtype=float)
    plevel = np.asarray(atm.level["pressure"], dtype=float) / atm.c.pconv
    tlayer = np.asarray(atm.layer["temperature"], dtype=float)
    player_cgs = np.asarray(atm.layer["pressure"], dtype=float)
    gravity = atm.planet.gravity / 100.0
    mmw = np.asarray(atm.layer["mmw"], dtype=float)
    colden = np.asarray(atm.layer["colden"], dtype=float)
    mix = atm.layer["mixingratios"]

    def x(m):
        return np.asarray(mix[m].values if hasattr(mix[m], "values") else mix[m], dtype=float)

    ACOEF = (tlayer / (tlevel[:-1] * tlevel[1:])) * (
        tlevel[1:] * plevel[1:] - tlevel[:-1] * plevel[:-1]) / (plevel[1:] - plevel[:-1])
    BCOEF = (tlayer / (tlevel[:-1] * tlevel[1:])) * (
        tlevel[:-1] - tlevel[1:]) / (plevel[1:] - plevel[:-1])
    COEF1 = atm.c.rgas * 273.15 ** 2 * .5E5 * (
        ACOEF * (plevel[1:] ** 2 - plevel[:-1] ** 2) + BCOEF * (
            2. / 3.) * (plevel[1:] ** 3 - plevel[:-1] ** 3)) / (
        1.01325 ** 2 * gravity * tlayer * mmw)
    cont_fac = []
    for m in atm.continuum_molecules:
        if m[0] == "H-" and m[1] == "bf":                  # optics.py:175-179
            cont_fac.append(x(m[0]) * colden / (mmw * atm.c.amu))
        elif m[0] == "H-" and m[1] == "ff":                # optics.py:187-193
            cont_fac.append(player_cgs * x("H") * np.asarray(atm.layer["electrons"]) * colden /
                            (tlayer * mmw * atm.c.amu * atm.c.k_b))
        elif m[0] == "H2-" and m[1] == "":                 # optics.py:203-213
            cont_fac.append(player_cgs * x("H2") * np.asarray(atm.layer["electrons"]) * colden /
                            (mmw * atm.c.amu))
        else:                                               # optics.py:224-227
            cont_fac.append(COEF1 * x(m[0]) * x(m[1]))
    if pl.get("premixed"):                                # optics.py:256-262
        mol_fac = [colden / mmw]
    else:
        mol_fac = [pl["fac"][i] * (colden * x(m) / mmw) for i, m in enumerate(pl["molecules"])]   # :246-249
    ray_names = [m for m in atm.rayleigh_molecules if m in opacityclass._ray]
    ray_fac = [colden * x(m) / mmw for m in ray_names]     # optics.py:265-271
    nfac = tlayer.shape[1] if tlayer.ndim == 2 else 0         # facet form of ATMSETUP: (nlayer, nfacets)

    def table(rows):
        """(nspecies, nlayer) -- or facet-major (nspecies, nfacets*nlayer) for a facet atmosphere"""
        if not rows:
            return np.zeros((0, nlayer * max(nfac, 1)))
        if nfac == 0:
            return np.array(rows).reshape((-1, nlayer))
        full = np.stack([np.broadcast_to(r, (nlayer, nfac)) for r in rows])       # (n, nlayer, nfac)
        return np.ascontiguousarray(full.transpose(0, 2, 1)).reshape(len(rows), nfac * nlayer)
    return table(mol_fac), table(cont_fac), ray_names, table(ray_fac)


def gas_stage(atm, opa, taugas, tauray, mix=None):
    """TAUGAS / TAURAY of one atmosphere into the given device planes (``k_opacity_gas``): table rows
    and weights from ``opa.get_opacities(atm)``, per-layer coefficients of reference optics.py:144-277.
    ``mix``: the argument tuple of the mixing (``compute_opacity_resident``) -- both stages then run as ONE launch
    and TAUGAS / TAURAY are not written."""
    pl = opa._plan
    nlayer, ngauss = atm.c.nlayer, opa.ngauss
    # per-layer coefficients: a plan shared by the wavelength blocks of one spectrum (picaso(devices=N)) computes them once
    fac = pl.get("_factors")
    if fac is None or fac[0] is not atm.layer["mixingratios"]:
        fac = (atm.layer["mixingratios"], _layer_factors(atm, opa))
        pl["_factors"] = fac
    mol_fac, cont_fac, ray_names, ray_fac = fac[1]
    cont_tabs = [opa._cia[p] for p in pl["cia_pairs"]]
    if pl.get("premixed"):
        mol_tabs, mol_mode = [pl.get("table", opa._kappa)], 2
        cont_rows = np.repeat(pl["cia_rows"][None], len(cont_tabs), axis=0) if cont_tabs else None
        cont_wts = np.repeat(pl["cia_wts"][None], len(cont_tabs), axis=0) if cont_tabs else None
    else:
        mol_tabs = [(opa._mol_log if opa.query_method == "linear" else opa._mol_raw)[m]
                    for m in pl["molecules"]]
        mol_mode = 1 if opa.query_method == "linear" else 0
        cont_rows = np.repeat(pl["cia_rows"][None], len(cont_tabs), axis=0) if cont_tabs else None
        cont_wts = None
    _gas_call(opa, nlayer, mol_tabs, pl["rows"] if mol_tabs else None,
              pl["wts"] if mol_tabs else None, mol_fac if mol_tabs else None, cont_tabs, cont_rows,
              cont_fac if cont_tabs else None, [opa._ray[m] for m in ray_names],
              ray_fac if ray_names else None, taugas, tauray, mol_mode=mol_mode, cont_wts=cont_wts,
              ngauss=ngauss, mix=mix)


def types_namespace_layer(atm_f, tlayer):
    """One facet's view of a facet-form atmosphere: what raman_plane_host reads."""
    import types
    return types.SimpleNamespace(c=atm_f.c, layer={"temperature": tlayer})


def raman_device(atm, opa, raman):
    """The Raman factor on the device as ``(DeviceArray or None, raman_rows)``: a ``(nlayer, nwno)`` plane
    (Oklopcic: depends on the layer temperatures, computed per call on the host) or one row of ``nwno`` values for
    every layer (Pollack: the table on the opacity grid, kept on the opacity object once formed)."""
    nlayer = atm.c.nlayer
    if raman == 1 and not _options().raman_planes:
        ref = os.environ.get("picaso_refdata")
        path = os.path.join(ref, "opacities", "raman_fortran.txt") if ref is not None else None
        st = os.stat(path) if path is not None and os.path.isfile(path) else None
        key = (path, st.st_mtime_ns, st.st_size) if st is not None else None
        hit = opa.__dict__.get("_raman_pollack")
        if hit is None or key is None or hit[0] != key:
            row = np.minimum(raman_pollack(1, 1e4 / opa.wno)[0], 0.99999)
            hit = opa.__dict__["_raman_pollack"] = (key, DeviceArray.from_host(np.ascontiguousarray(row), opa.ctx))
        return hit[1], 0
    if raman == 0 and not _options().raman_planes:
        out = DeviceArray((nlayer, opa.nwno), opa.ctx)
        raman_oklopcic_device(opa, np.asarray(atm.layer["temperature"], dtype=float), out)
        return out, nlayer
    rf = raman_plane_host(atm, opa, raman)
    return (DeviceArray.from_host(rf, opa.ctx), nlayer) if rf is not None else (None, nlayer)


def raman_oklopcic_device(opa, tlayer, out):
    """``min(compute_raman(...), 0.99999)`` (reference optics.py:285-294, 434-494) written into the device plane
    ``out`` ``(nlayer, nwno)`` by ``picaso_raman_oklopcic_dev``.  Everything that depends on the wavelength only --
    ``Q_i = c_i / wno**3 / (wno + deltanu_i)`` and ``Q_i * stellar_shifts[:, i]`` -- is formed once with the reference's
    numpy expressions and kept on the device with the opacity object (until ``raman_db`` or the
    ``raman_stellar_shifts`` array is replaced); per call only the 10 rotational populations of every layer
    (``j_fraction``) are computed on the host.  Same bits as the host function, which takes ~1 s per call at 1e5
    wavelengths."""
    db, shifts = opa.raman_db, opa.raman_stellar_shifts
    if db is None or shifts is None:
        raise Exception("raman='oklopcic' needs opa.raman_db (c, ji, deltanu) and opa.raman_stellar_shifts (nwno, "
                        "transitions): the ratio of the shifted to the unshifted stellar spectrum on the opacity grid")
    c, ji, dnu = (np.asarray(db[k], dtype=t) for k, t in (("c", float), ("ji", np.int32), ("deltanu", float)))
    hit = opa.__dict__.get("_raman_oklopcic")
    if hit is None or hit["shifts"] is not shifts or not (np.array_equal(hit["c"], c) and np.array_equal(hit["ji"], ji)
                                                          and np.array_equal(hit["dnu"], dnu)):
        wno, sh = np.asarray(opa.wno, dtype=float), np.asarray(shifts, dtype=float)
        Q = np.empty((c.size, wno.size))
        QS = np.zeros((c.size, wno.size))
        for i in range(c.size):
            Q[i] = c[i] / wno ** 3.0 / (wno + dnu[i])
            if dnu[i] != 0:
                QS[i] = Q[i] * sh[:, i]
        hit = opa.__dict__["_raman_oklopcic"] = dict(
            shifts=shifts, c=c.copy(), ji=np.ascontiguousarray(ji), dnu=dnu.copy(),
            isray=np.ascontiguousarray((dnu == 0).astype(np.int32)),
            Q=DeviceArray.from_host(Q, opa.ctx), QS=DeviceArray.from_host(QS, opa.ctx))
    tlayer = np.asarray(tlayer, dtype=float)
    nlayer = tlayer.size
    jat = np.ascontiguousarray(np.stack([j_fraction(j, tlayer) for j in range(10)]), dtype=np.float64)
    check(load().picaso_raman_oklopcic_dev(
        opa.ctx, _ci(nlayer), ctypes.c_long(opa.nwno), _ci(c.size), ptr(hit["Q"].addr), ptr(hit["QS"].addr),
        hit["ji"].ctypes.data_as(ctypes.c_void_p), hit["isray"].ctypes.data_as(ctypes.c_void_p), ptr(jat), _cd(0.99999),
        ptr(out.addr)), opa.ctx)


def raman_plane_host(atm, opa, raman):
    """Raman factor plane on the host (once per atmosphere, reference optics.py:285-306) or None."""
    if raman == 0:
        rf = compute_raman(opa.nwno, atm.c.nlayer, opa.wno, opa.raman_stellar_shifts,
                           np.asarray(atm.layer["temperature"], dtype=float), opa.raman_db["c"],
                           opa.raman_db["ji"], opa.raman_db["deltanu"])
        return np.minimum(rf, 0.99999)
    if raman == 1:
        return np.minimum(raman_pollack(atm.c.nlayer, 1e4 / opa.wno), 0.99999)      # optics.py:296-298
    return None


def raman_pollack(nlayer, wave, table=None):
    """Pollack+1986 Raman factor: the reference's tabulated ``raman_fortran.txt`` (two whitespace
    columns, wavelength in micron and factor) interpolated linearly onto ``wave`` and tiled over the
    layers (reference optics.py:584-652).  ``table`` = path or ``(w, f)`` arrays; by default the
    file is looked up where the reference looks, ``$picaso_refdata/opacities/raman_fortran.txt``."""
    if table is None:
        ref = os.environ.get("picaso_refdata")
        if ref is None:
            raise Exception("raman='pollack' reads $picaso_refdata/opacities/raman_fortran.txt: set the "
                            "picaso_refdata environment variable or use raman='oklopcic' / 'none'")
        table = os.path.join(ref, "opacities", "raman_fortran.txt")
    if isinstance(table, (str, bytes, os.PathLike)):
        if not os.path.isfile(table):
            raise Exception("raman='pollack': table %s not found" % table)
        try:                            # pandas' float parser, as the reference (optics.py:645): last bits differ from strtod's
            import pandas as pd
            dat = pd.read_csv(table, sep=r"\s+", header=None, names=["w", "f"])
            w, f = dat["w"].values, dat["f"].values
        except ImportError:
            w, f = np.loadtxt(table, unpack=True)
    else:
        w, f = (np.asarray(x, dtype=float) for x in table)
    row = np.interp(np.asarray(wave, dtype=float), w, f)
    return np.repeat(row[None, :], nlayer, axis=0)


def gas_stage_facets(atm_f, opa, nfac, tg3, tr3, exclude_mol=1):
    """TAUGAS / TAURAY of every facet from ONE facet-form atmosphere (``ATMSETUP`` with
    ``(nlevel, nfacets)`` columns): the facet stack ``(nfacets, nlayer, nwno)`` is a tall atmosphere of
    ``nfacets*nlayer`` layers for ``k_opacity_gas``, so the bracket search, the layer coefficients and the
    launch run once over all facets (chunked only so the per-layer tables fit one upload slot) instead
    of once per facet (reference loop: justdoit.py:437-471).  Same arithmetic per element as the
    per-facet path: the results are bit-identical (tests/test_ck_optics.py)."""
    import types
    nlayer = atm_f.c.nlayer
    ntot = nfac * nlayer

    def flat(a):                      # (nlayer, nfacets | 1) -> facet-major (nfacets*nlayer,)
        return np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=float), (nlayer, nfac)).T).ravel()
    tall = types.SimpleNamespace(
        c=types.SimpleNamespace(nlayer=ntot, pconv=atm_f.c.pconv),
        layer={"temperature": flat(atm_f.layer["temperature"]), "pressure": flat(atm_f.layer["pressure"])},
        molecules=atm_f.molecules, continuum_molecules=atm_f.continuum_molecules)
    fast = getattr(atm_f, "_fast_tall", None)
    if fast is not None and fast[2] is opa and exclude_mol == 1:       # formed with the atmosphere (fastsetup.setup_facets)
        opa._plan = pl = fast[0]
        opa.molecular_opa, opa.continuum_opa = _LazyPlanes(opa, "mol"), _LazyPlanes(opa, "cia")
        mol_fac, cont_fac, ray_names, ray_fac = fast[1]
    else:
        opa.get_opacities(tall, exclude_mol=exclude_mol)
        pl = opa._plan
        if pl.get("premixed"):
            raise Exception("the 3-D path takes monochromatic opacities")
        mol_fac, cont_fac, ray_names, ray_fac = _layer_factors(atm_f, opa)
    mol_tabs = [(opa._mol_log if opa.query_method == "linear" else opa._mol_raw)[m] for m in pl["molecules"]]
    mol_mode = 1 if opa.query_method == "linear" else 0
    cont_tabs = [opa._cia[p] for p in pl["cia_pairs"]]
    ray_tabs = [opa._ray[m] for m in ray_names]
    per_layer = len(mol_tabs) * (16 + 32 + 8) + len(cont_tabs) * (4 + 8) + len(ray_tabs) * 8 + 8
    fchunk = max(1, int((3600 * 1024) // (per_layer * nlayer)))         # one 4 MB upload slot per launch (common.hpp SLOT_BYTES)
    for f0 in range(0, nfac, fchunk):
        f1 = min(nfac, f0 + fchunk)
        sl = slice(f0 * nlayer, f1 * nlayer)
        cont_rows = np.repeat(pl["cia_rows"][None, sl], len(cont_tabs), axis=0) if cont_tabs else None
        _gas_call(opa, (f1 - f0) * nlayer, mol_tabs, pl["rows"][:, sl] if mol_tabs else None,
                  pl["wts"][:, sl] if mol_tabs else None, mol_fac[:, sl] if mol_tabs else None, cont_tabs,
                  cont_rows, cont_fac[:, sl] if cont_tabs else None, ray_tabs, ray_fac[:, sl] if ray_tabs else None,
                  tg3.row_block(f0), tr3.row_block(f0), mol_mode=mol_mode, ngauss=1)


def compute_opacity_facets(atms, opacityclass, numg, numt, stream=2, delta_eddington=True, test_mode=None,
                           raman=0, clouds_3d=None, exclude_mol=1, want=None):
    """3-D path: the 13 ``(nlayer|nlevel, nwno, numg, numt)`` planes ``get_reflected_3d`` /
    ``get_thermal_3d`` take, from one atmosphere per facet (``atms[g][t]``; reference
    justdoit.py:444-471).  Gas + Rayleigh optical depths are gathered facet by facet into a
    facet-major stack; one launch then mixes all facets (cloud inputs ``(nlayer, nwno, numg, numt)``
    arrays ``opd``/``w0``/``g0`` or None) and writes the planes with the facet index fastest."""
    opa = opacityclass
    ctx = opa.ctx
    nfac = numg * numt
    nlayer, nwno = (atms[0][0] if isinstance(atms, list) else atms).c.nlayer, opa.nwno
    if opa.ngauss != 1:
        raise Exception("compute_opacity_facets takes monochromatic opacities")
    tg3, tr3 = DeviceArray((nfac, nlayer, nwno), ctx), DeviceArray((nfac, nlayer, nwno), ctx)
    on_dev = not _options().raman_planes
    row_mode = raman == 1 and on_dev
    rf3 = [] if raman in (0, 1) and not on_dev else None      # host planes, one per facet
    d_rf3 = DeviceArray((nfac, nlayer, nwno), ctx) if raman == 0 and on_dev else None   # Oklopcic: layer temperatures
    if not isinstance(atms, list):                            # one facet-form atmosphere: batched gas stage
        atm_f = atms
        gas_stage_facets(atm_f, opa, nfac, tg3, tr3, exclude_mol=exclude_mol)
        if rf3 is not None:
            tl = np.broadcast_to(np.asarray(atm_f.layer["temperature"], dtype=float), (nlayer, nfac))
            for f in range(nfac):
                one = types_namespace_layer(atm_f, tl[:, f])
                rf3.append(raman_plane_host(one, opa, raman))
        if d_rf3 is not None:
            tl = np.broadcast_to(np.asarray(atm_f.layer["temperature"], dtype=float), (nlayer, nfac))
            # all facets in one launch: the tall atmosphere's nfac * nlayer layer temperatures, facet-major
            raman_oklopcic_device(opa, np.ascontiguousarray(tl.T).ravel(), d_rf3)
        atms = None
    for g in range(numg if atms is not None else 0):
        for t in range(numt):
            f = g * numt + t
            opa.get_opacities(atms[g][t], exclude_mol=exclude_mol)
            gas_stage(atms[g][t], opa, tg3.row_block(f), tr3.row_block(f))
            if rf3 is not None:
                rf3.append(raman_plane_host(atms[g][t], opa, raman))
            if d_rf3 is not None:
                raman_oklopcic_device(opa, atms[g][t].layer["temperature"], d_rf3.row_block(f))
    d_rf, rf_rows = (DeviceArray.from_host(np.stack(rf3), ctx) if rf3 else d_rf3), nlayer
    if row_mode:                                              # Pollack: one row for every layer and facet
        d_rf, rf_rows = raman_device(atms[0][0] if isinstance(atms, list) else atm_f, opa, raman)
    d_c = [None, None, None]
    if clouds_3d is not None:
        from .device import broadcast_facets, regrid_facets, regrid_rows
        d_c = []
        in_wno = clouds_3d.get("wavenumber") if isinstance(clouds_3d, dict) else None
        if in_wno is not None and len(in_wno) == nwno and np.array_equal(in_wno, opa.wno):
            in_wno = None
        if in_wno is not None:
            # tables on a wavenumber grid of their own (virga's 196 points): regridded on the device with numpy.interp's
            # bits -- the reference's per-facet get_clouds -> wavelength.regrid (atmsetup.py:609-622) -- instead of
            # (nlayer, nwno, nfacets) host arrays: 3 x 576 MB and seconds of numpy at 1e5 wavelengths x 64 facets
            in_wno = np.asarray(in_wno, dtype=np.float64)
            nin = in_wno.size
            order = np.argsort(in_wno, kind="stable") if np.any(np.diff(in_wno) < 0) else None
            d_x = _wno_device(opa, opa.wno)
        if in_wno is not None and nin >= 2:
            # the three tables in ONE upload and ONE launch: (3 nlayer[, nfacets], nin) rows, kept on the cloud dictionary
            # while its arrays are the same objects (a spectrum() called again, the phases of a curve that share a map)
            arrs = [clouds_3d[k] for k in ("opd", "w0", "g0")]
            memo = _cloud_memo_get("rows", clouds_3d)
            stamp = _table_fingerprint(arrs, clouds_3d["wavenumber"])
            if memo is None or memo[1] != stamp:
                tabs = [np.asarray(x, dtype=float) for x in arrs]
                shared = all(t.size == nlayer * nin for t in tabs)
                if shared:
                    rows = np.concatenate([t.reshape(nlayer, nin) for t in tabs])
                    rows = rows if order is None else rows[:, order]
                else:
                    rows = np.concatenate([np.moveaxis(np.broadcast_to(t.reshape(nlayer, nin, -1), (nlayer, nin, nfac)), 1, 2)
                                           for t in tabs])
                    rows = np.ascontiguousarray(rows if order is None else rows[:, :, order])
                memo = (arrs, stamp, shared, np.ascontiguousarray(rows),
                        np.ascontiguousarray(in_wno if order is None else in_wno[order]), {})
                _cloud_memo_put("rows", clouds_3d, memo)
            _, _, shared, rows, xp, resident_tabs = memo
            key = (os.getpid(), getattr(ctx, "value", ctx))
            if key not in resident_tabs:         # the compact tables stay in HBM with the dictionary (27 MB at 64 facets)
                resident_tabs[key] = (DeviceArray.from_host(xp, ctx), DeviceArray.from_host(rows, ctx))
            xp, rows = resident_tabs[key]
            if shared:
                d_all = regrid_rows(xp, rows, d_x, ctx)                                  # (3 nlayer, nwno)
                d_c = [broadcast_facets(d_all.row_range(j * nlayer, (j + 1) * nlayer), nfac) for j in range(3)]
            else:
                d_all = regrid_facets(xp, rows, d_x, ctx)                                # (3 nlayer, nwno, nfacets)
                d_c = [d_all.row_range(j * nlayer, (j + 1) * nlayer) for j in range(3)]
            for d in d_c:
                d._inputs = d_all
        else:
            for k in ("opd", "w0", "g0"):
                a = np.asarray(clouds_3d[k], dtype=float)
                if in_wno is not None:               # a single wavenumber: the value everywhere
                    a = np.repeat(a.reshape(nlayer, 1, -1), nwno, axis=1)
                    d_c.append(broadcast_facets(DeviceArray.from_host(np.ascontiguousarray(a[:, :, 0]), ctx), nfac)
                               if a.shape[2] == 1 else DeviceArray.from_host(np.ascontiguousarray(a), ctx))
                elif a.size == nlayer * nwno:        # one table for the whole disk: tiled over the facets on the device
                    d_c.append(broadcast_facets(DeviceArray.from_host(a.reshape(nlayer, nwno), ctx), nfac))
                else:
                    d_c.append(DeviceArray.from_host(a.reshape(nlayer, nwno, nfac), ctx))
    tm = 0
    if test_mode is not None:
        tm = 1 if test_mode == "rayleigh" else 2
    # only the planes the requested legs read are allocated and written (each is nfacets x 72 MB at
    # 1e5 wavelengths x 90 layers): 11 for reflected light, 3 for thermal emission
    out = {k: (DeviceArray(((nlayer + 1 if k in ("tau", "tau_og") else nlayer), nwno, numg, numt), ctx)
               if (want is None or k in want) else None) for k in OUT_NAMES}
    check(load().picaso_compute_opacity_facets_dev(
        ctx, _ci(nlayer), _ci(nwno), _ci(nfac), ptr(tg3.addr), ptr(tr3.addr),
        *[ptr(d.addr) if d is not None else None for d in d_c], ptr(d_rf.addr) if d_rf is not None else None,
        _ci(rf_rows), _cd(0.99999), _ci(tm), _ci(1 if delta_eddington else 0), _ci(stream),
        *[ptr(out[k].addr) if out[k] is not None else None for k in OUT_NAMES]), ctx)
    return {k: v for k, v in out.items() if v is not None}


try:                                   # in the image; the fallback is hashlib (slower, same coverage)
    from xxhash import xxh3_128 as _xxh3
except ImportError:                    # pragma: no cover
    _xxh3 = None

_CLOUD_MEMO = {}      # (kind, id(cloud dictionary)) -> memo; the caller's dictionary is never written to (it may be deep-copied)


def _cloud_memo_get(kind, clouds_3d):
    return _CLOUD_MEMO.get((kind, id(clouds_3d)))


def _cloud_memo_put(kind, clouds_3d, memo):
    if len(_CLOUD_MEMO) >= 24:                  # a phase curve's maps; older ones go (their device tables with them)
        for k in list(_CLOUD_MEMO)[:8]:
            del _CLOUD_MEMO[k]
    _CLOUD_MEMO[(kind, id(clouds_3d))] = memo


_DIGEST_POOL = None
_DIGEST_PARALLEL_BYTES = 4 << 20


def content_digest(a):
    """128-bit digest of EVERY byte of ``a`` (plus shape and dtype): the key of the content caches.  The reference
    re-reads its inputs on every call (justdoit.py:437-449, atmsetup.py:609-622, deq_chem.py:334-384), so a device copy
    may be reused only while the host array is bit for bit the one it was made from; an edit of a single element in
    place changes the digest.  xxh3 runs at memory speed -- 45 GB/s on one core of the MI355X box's host, 100 GB/s with
    the four chunks of a large array hashed side by side (the extension releases the GIL): 0.3 ms for the 27 MB of three
    (90, 196, 64) cloud tables; blake2b is the fallback."""
    global _DIGEST_POOL
    v = np.ascontiguousarray(a)
    head = ("%s|%s|" % (v.dtype.str, v.shape)).encode()
    buf = memoryview(v.reshape(-1).view(np.uint8)) if v.size else b""
    if _xxh3 is not None:
        h = _xxh3(head)
        n = len(buf)
        if n >= _DIGEST_PARALLEL_BYTES:
            if _DIGEST_POOL is None or _DIGEST_POOL[0] != os.getpid():      # (threads do not survive a fork)
                from concurrent.futures import ThreadPoolExecutor
                _DIGEST_POOL = (os.getpid(), ThreadPoolExecutor(4, thread_name_prefix="picaso_amd_digest"))
            cuts = [(i * n // 4) & ~63 for i in range(4)] + [n]
            for part in _DIGEST_POOL[1].map(lambda i: _xxh3(buf[cuts[i]:cuts[i + 1]]).digest(), range(4)):
                h.update(part)
        else:
            h.update(buf)
        return h.digest()
    import hashlib
    h = hashlib.blake2b(head, digest_size=16)
    h.update(buf)
    return h.digest()


def _table_fingerprint(arrs, wavenumber):
    """Identity of the table objects plus a digest of ALL their contents: a memo kept with a cloud dictionary is dropped
    when an array is replaced OR edited in place anywhere (scaled, one (layer, facet) row rewritten, one element
    changed).  The ids also guard the memo against a dictionary id reused after collection."""
    return tuple((id(a), content_digest(a)) for a in list(arrs) + [wavenumber])


def _facet_major_cloud_tables(clouds_3d, nlayer, nfac, ctx, stamp=None):
    """Cloud tables on their own increasing wavenumber grid as resident ``(3, nfacets * nlayer, nin)`` rows in
    facet-major order (opd, w0, g0) with their grid, kept on the cloud dictionary while its arrays are the same
    objects.  None when the tables are not of that kind."""
    in_wno = clouds_3d.get("wavenumber") if isinstance(clouds_3d, dict) else None
    if in_wno is None or np.size(in_wno) < 2:
        return None
    in_wno = np.asarray(in_wno, dtype=np.float64)
    nin = in_wno.size
    arrs = [clouds_3d[k] for k in ("opd", "w0", "g0")]
    if any(np.size(a) not in (nlayer * nin, nlayer * nin * nfac) for a in arrs):
        return None
    memo = _cloud_memo_get("tall", clouds_3d)
    if stamp is None:               # (a caller that asks once per wavelength block digests the tables once and hands it in)
        stamp = _table_fingerprint(arrs, clouds_3d["wavenumber"])
    if memo is None or memo[1] != stamp:
        order = np.argsort(in_wno, kind="stable") if np.any(np.diff(in_wno) < 0) else slice(None)
        tall = np.stack([np.moveaxis(np.broadcast_to(np.asarray(a, dtype=float).reshape(nlayer, nin, -1), (nlayer, nin, nfac)),
                                     (0, 1, 2), (1, 2, 0))[:, :, order].reshape(nfac * nlayer, nin) for a in arrs])
        memo = (arrs, stamp, np.ascontiguousarray(tall), np.ascontiguousarray(in_wno[order]), {})
        _cloud_memo_put("tall", clouds_3d, memo)
    key = (os.getpid(), getattr(ctx, "value", ctx))
    if key not in memo[4]:
        memo[4][key] = (DeviceArray.from_host(memo[3], ctx), DeviceArray.from_host(memo[2], ctx))
    return memo[4][key] + (memo[2], nin)


def tall_plan(atm_f, opa, nfac, exclude_mol=1):
    """Table rows / weights (``opa._plan``) and per-layer coefficients (``_layer_factors``) of the TALL atmosphere of a
    3-D spectrum: the ``nfac * nlayer`` layers of all facets, facet-major.  From ``picaso_host_setup_facets`` when the
    facet-form set-up went through it (``atm_f._fast_tall``), else with the mirror's table search on the flattened
    layer temperatures and pressures."""
    nlayer = atm_f.c.nlayer
    fast = getattr(atm_f, "_fast_tall", None)
    if fast is not None and fast[2] is opa and exclude_mol == 1:
        opa._plan = fast[0]
        opa.molecular_opa, opa.continuum_opa = _LazyPlanes(opa, "mol"), _LazyPlanes(opa, "cia")
        return fast[0], fast[1]
    import types

    def flat(a):
        return np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=float), (nlayer, nfac)).T).ravel()
    tall = types.SimpleNamespace(
        c=types.SimpleNamespace(nlayer=nfac * nlayer, pconv=atm_f.c.pconv),
        layer={"temperature": flat(atm_f.layer["temperature"]), "pressure": flat(atm_f.layer["pressure"])},
        molecules=atm_f.molecules, continuum_molecules=atm_f.continuum_molecules)
    opa.get_opacities(tall, exclude_mol=exclude_mol)
    return opa._plan, _layer_factors(atm_f, opa)


def compute_opacity_facet_major(atm_f, opacityclass, numg, numt, stream=2, delta_eddington=True, raman=2, exclude_mol=1,
                                want=("dtau", "w0"), cloud_tables=None):
    """The planes of a 3-D spectrum WITHOUT cloud in facet-major layout ``(nfacets, nlayer, nwno)``, from one fused
    gas + mixing launch over the tall atmosphere of all facets (``picaso_gas_compute_opacity_dev`` with
    ``nfacets * nlayer`` layers): no TAUGAS / TAURAY stack in HBM and no transposing mixing launch
    (``compute_opacity_facets``: 0.76 + 0.79 ms at 64 facets x 90 layers x 12 500 wavelengths; this: 0.9).  Only the
    planes a cloud-free column cannot re-derive exist (``want`` out of dtau, w0, w0_no_raman); the solvers take them
    through ``resident.reflected_3d_fm_batch / thermal_3d_fm_batch``.  Same arithmetic per element: same bits.
    ``cloud_tables`` (``_facet_major_cloud_tables``): cloud tables on their own wavenumber grid, interpolated inside the
    launch like the 1-D path's (numpy.interp's bits) -- no TAUGAS / TAURAY stack, no regridded or tiled cloud planes."""
    opa = opacityclass
    ctx = opa.ctx
    nfac = numg * numt
    nlayer, nwno = atm_f.c.nlayer, opa.nwno
    ntot = nfac * nlayer
    pl, (mol_fac, cont_fac, ray_names, ray_fac) = tall_plan(atm_f, opa, nfac, exclude_mol)
    mol_tabs = [(opa._mol_log if opa.query_method == "linear" else opa._mol_raw)[m] for m in pl["molecules"]]
    mol_mode = 1 if opa.query_method == "linear" else 0
    cont_tabs = [opa._cia[p] for p in pl["cia_pairs"]]
    ray_tabs = [opa._ray[m] for m in ray_names]
    # the Raman factor: none (the constant), Pollack's row for every layer and facet, or Oklopcic's plane per facet
    d_rf, rf_rows = None, 0
    if raman == 1:
        d_rf, rf_rows = raman_device(atm_f, opa, 1)
    elif raman == 0:
        d_rf = DeviceArray((nfac, nlayer, nwno), ctx)
        tl = np.broadcast_to(np.asarray(atm_f.layer["temperature"], dtype=float), (nlayer, nfac))
        raman_oklopcic_device(opa, np.ascontiguousarray(tl.T).ravel(), d_rf)      # all facets in one launch
    out = {k: DeviceArray((nfac, nlayer, nwno), ctx) for k in OUT_NAMES if k in want}
    per_layer = len(mol_tabs) * (16 + 32 + 8) + len(cont_tabs) * (4 + 8) + len(ray_tabs) * 8 + 8
    fchunk = max(1, int((3600 * 1024) // (per_layer * nlayer)))         # per-layer tables of a launch: one 4 MB slot
    for f0 in range(0, nfac, fchunk):
        f1 = min(nfac, f0 + fchunk)
        sl = slice(f0 * nlayer, f1 * nlayer)
        nl_c = (f1 - f0) * nlayer
        off = f0 * nlayer * nwno * 8
        cld_tab, keep = (_ci(0), None, None, None), None
        if cloud_tables is not None:
            d_xp, d_tall, h_tall, nin = cloud_tables
            if f0 == 0 and f1 == nfac:           # one launch: the resident (3 ntot, nin) rows as they are
                d_fp = d_tall
            else:                                # this chunk's opd / w0 / g0 rows, contiguous
                d_fp = keep = DeviceArray.from_host(np.ascontiguousarray(h_tall[:, sl].reshape(3 * nl_c, nin)), ctx)
            cld_tab = (_ci(nin), ptr(d_xp.addr), ptr(d_fp.addr), ptr(_wno_device(opa, opa.wno).addr))
        mix = (None, None, None,
               ptr(d_rf.addr + (off if rf_rows else 0)) if d_rf is not None else None, _ci(nl_c if (d_rf is not None and raman == 0) else 0),
               _cd(0.99999), _ci(0), _ci(1 if delta_eddington else 0), _ci(stream),
               *[ptr(out[k].addr + off) if k in out else None for k in OUT_NAMES], _ci(0), *cld_tab)
        if keep is not None:
            out.setdefault("_cloud_chunks", []).append(keep)
        cont_rows = np.repeat(pl["cia_rows"][None, sl], len(cont_tabs), axis=0) if cont_tabs else None
        _gas_call(opa, nl_c, mol_tabs, pl["rows"][:, sl] if mol_tabs else None, pl["wts"][:, sl] if mol_tabs else None,
                  mol_fac[:, sl] if mol_tabs else None, cont_tabs, cont_rows, cont_fac[:, sl] if cont_tabs else None,
                  ray_tabs, ray_fac[:, sl] if ray_tabs else None, None, None, mol_mode=mol_mode, ngauss=1, mix=mix)
    out["_fm"] = True
    if d_rf is not None:
        out["_raman"] = d_rf              # the launch is asynchronous: its inputs live as long as its outputs
    if cloud_tables is not None:
        out["_cloud_tables"] = cloud_tables[:2]
    return out


def _cloud_planes_facet_major(clouds_3d, opa, nlayer, nfac, ctx):
    """The cloud tables of the 3-D path as resident facet-major planes ``(d_opd, d_w0, d_g0, stride)``, each
    ``(nfacets | 1, nlayer, nwno)``; ``stride`` = elements between two facets (0: one set for the whole disk).  Tables on
    a wavenumber grid of their own are regridded on the device with numpy.interp's bits -- the reference's per-facet
    get_clouds -> wavelength.regrid (atmsetup.py:609-622) -- from the compact rows kept with the cloud dictionary
    (``_facet_major_cloud_tables``: re-uploaded when any byte of the caller's arrays changes)."""
    nwno = opa.nwno
    in_wno = clouds_3d.get("wavenumber") if isinstance(clouds_3d, dict) else None
    if in_wno is not None and np.size(in_wno) == nwno and np.array_equal(in_wno, opa.wno):
        in_wno = None
    if in_wno is not None and np.size(in_wno) >= 2:
        tabs = _facet_major_cloud_tables(clouds_3d, nlayer, nfac, ctx)
        if tabs is None:
            raise Exception("clouds_3d: opd / w0 / g0 must hold nlayer x nwavenumber (x nfacets) numbers")
        d_xp, d_tall, _, nin = tabs
        rows = regrid_rows(d_xp, d_tall.reshape((3 * nfac * nlayer, nin)), _wno_device(opa, opa.wno), ctx)
        rows._inputs = (d_xp, d_tall)
        r3 = rows.reshape((3, nfac * nlayer, nwno))
        return r3.row_block(0), r3.row_block(1), r3.row_block(2), nlayer * nwno, rows
    out = []
    shared = True
    for k in ("opd", "w0", "g0"):
        a = np.asarray(clouds_3d[k], dtype=float)
        if in_wno is not None:                   # a single wavenumber: the value everywhere
            a = np.repeat(a.reshape(nlayer, 1, -1), nwno, axis=1)
        if a.size == nlayer * nwno:
            out.append(a.reshape(1, nlayer, nwno))
        else:
            shared = False
            out.append(np.moveaxis(a.reshape(nlayer, nwno, nfac), 2, 0))
    if not shared:
        out = [np.broadcast_to(a, (nfac, nlayer, nwno)) for a in out]
    devs = [DeviceArray.from_host(np.ascontiguousarray(a), ctx) for a in out]
    return devs[0], devs[1], devs[2], (0 if shared else nlayer * nwno), None


def compute_opacity_facet_major_ck(atm_f, opacityclass, numg, numt, stream=2, delta_eddington=True, test_mode=None,
                                   raman=2, clouds_3d=None, exclude_mol=1, want=None):
    """The planes of a 3-D spectrum on CORRELATED-K tables (reference justdoit.py:437-471 with ``ngauss > 1``:
    ``get_opacities`` + ``compute_opacity`` facet by facet, ``DTAU_3d[:, :, g, t, :] = dtau``) in facet-major layout
    ``(nfacets, nlayer|nlevel, nwno, ngauss)``: ONE gas launch over the tall atmosphere of all facets
    (``nfacets * nlayer`` layers; premixed table rows or, for on-the-fly mixing, one resort-rebin launch over all of
    them), then ``picaso_compute_opacity_facet_major_ck_dev``.  Each facet's numbers are those of
    ``compute_opacity_resident`` on that facet's atmosphere (tests/test_ck3d_gpu.py).  ``want``: the planes to write
    (default all 13).  The solvers: ``resident.reflected_3d_ck`` / ``thermal_3d_ck``."""
    import types
    opa = opacityclass
    ctx = opa.ctx
    nfac = numg * numt
    nlayer, nwno, ngauss = atm_f.c.nlayer, opa.nwno, opa.ngauss
    ntot = nfac * nlayer

    def flat(a):                      # (nlayer, nfacets | 1) -> facet-major (nfacets*nlayer,)
        a = np.asarray(a, dtype=float)
        return np.ascontiguousarray(np.broadcast_to(a.reshape(nlayer, -1), (nlayer, nfac)).T).ravel()
    mix = atm_f.layer["mixingratios"]
    tall = types.SimpleNamespace(
        c=types.SimpleNamespace(nlayer=ntot, pconv=atm_f.c.pconv),
        layer={"temperature": flat(atm_f.layer["temperature"]), "pressure": flat(atm_f.layer["pressure"]),
               "mixingratios": {m: flat(mix[m].values if hasattr(mix[m], "values") else mix[m]) for m in atm_f.molecules}},
        molecules=atm_f.molecules, continuum_molecules=atm_f.continuum_molecules)
    fast = getattr(atm_f, "_fast_tall", None)
    if fast is not None and fast[2] is opa and fast[0].get("premixed") and exclude_mol == 1:
        # the tall plan and coefficients came with the facet-form set-up (fastsetup.setup_facets)
        opa._plan = pl = fast[0]
        opa.continuum_opa, opa.molecular_opa = _LazyPlanes(opa, "cia"), None
        mol_fac, cont_fac, ray_names, ray_fac = fast[1]
    else:
        opa.get_opacities(tall, exclude_mol=exclude_mol)
        pl = opa._plan
        mol_fac, cont_fac, ray_names, ray_fac = _layer_factors(atm_f, opa)
    cont_tabs = [opa._cia[p] for p in pl["cia_pairs"]]
    ray_tabs = [opa._ray[m] for m in ray_names]
    mol_tabs = [pl.get("table", opa._kappa)]
    taugas, tauray = DeviceArray((nfac, nlayer, nwno, ngauss), ctx), DeviceArray((nfac, nlayer, nwno), ctx)
    per_layer = (16 + 32 + 8) + len(cont_tabs) * (8 + 16 + 8) + len(ray_tabs) * 8 + 8
    fchunk = max(1, int((3600 * 1024) // (per_layer * nlayer)))         # per-layer tables of a launch: one 4 MB slot
    for f0 in range(0, nfac, fchunk):
        f1 = min(nfac, f0 + fchunk)
        sl = slice(f0 * nlayer, f1 * nlayer)
        cont_rows = np.repeat(pl["cia_rows"][None, sl], len(cont_tabs), axis=0) if cont_tabs else None
        cont_wts = np.repeat(pl["cia_wts"][None, sl], len(cont_tabs), axis=0) if cont_tabs else None
        _gas_call(opa, (f1 - f0) * nlayer, mol_tabs, pl["rows"][:, sl], pl["wts"][:, sl], mol_fac[:, sl], cont_tabs,
                  cont_rows, cont_fac[:, sl] if cont_tabs else None, ray_tabs, ray_fac[:, sl] if ray_tabs else None,
                  taugas.row_block(f0), tauray.row_block(f0), mol_mode=2, cont_wts=cont_wts, ngauss=ngauss)
    # the Raman factor: none (the constant), Pollack's row for every layer and facet, or Oklopcic's plane per facet
    d_rf, rf_rows = None, 0
    if raman == 1:
        d_rf, rf_rows = raman_device(atm_f, opa, 1)
    elif raman == 0:
        d_rf, rf_rows = DeviceArray((nfac, nlayer, nwno), ctx), nlayer
        tl = np.broadcast_to(np.asarray(atm_f.layer["temperature"], dtype=float).reshape(nlayer, -1), (nlayer, nfac))
        raman_oklopcic_device(opa, np.ascontiguousarray(tl.T).ravel(), d_rf)      # all facets in one launch
    d_cld = (None, None, None, 0, None)
    if clouds_3d is not None:
        d_cld = _cloud_planes_facet_major(clouds_3d, opa, nlayer, nfac, ctx)
    tm = 0
    if test_mode is not None:
        tm = 1 if test_mode == "rayleigh" else 2
    out = {k: (DeviceArray((nfac, (nlayer + 1 if k in ("tau", "tau_og") else nlayer), nwno, ngauss), ctx)
               if (want is None or k in want) else None) for k in OUT_NAMES}
    check(load().picaso_compute_opacity_facet_major_ck_dev(
        ctx, _ci(nfac), _ci(nlayer), _ci(nwno), _ci(ngauss), ptr(taugas.addr), ptr(tauray.addr),
        *[ptr(d.addr) if d is not None else None for d in d_cld[:3]], ctypes.c_long(d_cld[3]),
        ptr(d_rf.addr) if d_rf is not None else None, _ci(rf_rows), _cd(0.99999), _ci(tm),
        _ci(1 if delta_eddington else 0), _ci(stream),
        *[ptr(out[k].addr) if out[k] is not None else None for k in OUT_NAMES]), ctx)
    out = {k: v for k, v in out.items() if v is not None}
    out["_fm"], out["_ck"] = True, True
    out["_inputs"] = (taugas, tauray, d_rf, d_cld, pl.get("table"))     # asynchronous launch: inputs live with the outputs
    return out


def _wno_device(opa, wno):
    """The wavenumber grid as a device vector; the opacity object's own grid is uploaded once."""
    if wno is opa.wno:
        hit = opa.__dict__.get("_d_wno")
        if hit is None:
            hit = opa.__dict__["_d_wno"] = DeviceArray.from_host(np.ascontiguousarray(wno, dtype=np.float64), opa.ctx)
        return hit
    return DeviceArray.from_host(np.ascontiguousarray(wno, dtype=np.float64), opa.ctx)


@_lib.serialized
def compute_opacity_resident(atmosphere, opacityclass, ngauss=1, stream=2, delta_eddington=True,
                             test_mode=None, raman=0, fthin_cld=None, do_holes=False,
                             full_output=False, want=None):
    """GPU ``compute_opacity`` returning a dict of the 13 planes as DeviceArrays: ``(rows, nwno)``
    for monochromatic opacities, ``(rows, nwno, ngauss)`` (reference layout, optics.py:423-431)
    for correlated-k tables.  (3-D path: ``compute_opacity_facets``.)  ``want``: names of the planes
    the caller will read (default all 13); the others are neither allocated nor written."""
    atm, opa = atmosphere, opacityclass
    if ngauss != opa.ngauss:
        raise Exception("compute_opacity: ngauss=%d but the opacity tables have %d Gauss points"
                        % (ngauss, opa.ngauss))
    ctx = opa.ctx
    nlayer, nwno = atm.c.nlayer, opa.nwno
    if opa._plan is None or opa._plan["nlayer"] != nlayer:
        raise Exception("call opacityclass.get_opacities(atmosphere) first")
    gshape = (nwno,) if ngauss == 1 else (nwno, ngauss)
    # Monochromatic tables: gas stage and mixing as ONE launch (picaso_gas_compute_opacity_dev; TAUGAS / TAURAY stay in
    # registers) unless the caller wants those two planes back (full_output) or a level plane without its layer plane.
    # PICASO_AMD_UNFUSED_OPACITY=1: the two launches (A/B, tests) -- same bits either way.
    fused = (ngauss == 1 and not full_output and not _options().unfused_opacity
             and (want is None or (("tau" not in want or "dtau" in want) and ("tau_og" not in want or "dtau_og" in want))))
    if not fused:
        taugas, tauray = DeviceArray((nlayer,) + gshape, ctx), DeviceArray((nlayer, nwno), ctx)
        gas_stage(atm, opa, taugas, tauray)
    (raman_plane, raman_rows), raman_const = raman_device(atm, opa, raman), 0.99999
    cld = atm.layer["cloud"]

    def plane(x):       # (nlayer, nwno) float64 without a copy when the caller already has one
        a = np.asarray(x, dtype=float)
        return a if a.shape == (nlayer, nwno) else np.ascontiguousarray(np.broadcast_to(a, (nlayer, nwno)))

    def taucld_host():
        t = plane(cld["opd"])
        return fthin_cld * t if do_holes else t             # optics.py:314-315
    on_device = isinstance(cld, CloudTables) and np.size(cld.wno) == nwno and not _options().host_regrid
    cld_tab = (_ci(0), None, None, None)
    tab_keep = None
    if getattr(atm, "cloud_free", False) and not do_holes:  # no cloud profile: NULL planes read as zero
        d_cld = d_w0 = d_g0 = None
    elif on_device and fused and not do_holes and not _options().regrid_planes:
        # tables on their own wavenumber grid, interpolated INSIDE the fused opacity launch (numpy.interp's bits, as
        # picaso_regrid_rows_dev): no regridded planes in HBM
        stack = cld.__dict__.get("_stack")
        if stack is None:
            stack = cld.__dict__["_stack"] = np.concatenate([cld.compact[k] for k in ("opd", "w0", "g0")])
        tab_keep = (DeviceArray.from_host(np.ascontiguousarray(cld.in_wno, dtype=np.float64), ctx),
                    DeviceArray.from_host(np.ascontiguousarray(stack, dtype=np.float64), ctx), _wno_device(opa, cld.wno))
        cld_tab = (_ci(int(np.size(cld.in_wno))), ptr(tab_keep[0].addr), ptr(tab_keep[1].addr), ptr(tab_keep[2].addr))
        d_cld = d_w0 = d_g0 = None
    elif on_device:     # tables on their own wavenumber grid: interpolated where they are used (same bits)
        d_x = _wno_device(opa, cld.wno)
        if do_holes:    # the thinned optical depth has its own factor (optics.py:314-315)
            d_cld = regrid_rows(cld.in_wno, cld.compact["opd"], d_x, ctx, scale=fthin_cld)
            both = regrid_rows(cld.in_wno, np.concatenate([cld.compact["w0"], cld.compact["g0"]]), d_x, ctx)
            both3 = both.reshape((2, nlayer, nwno))
            d_w0, d_g0 = both3.row_block(0), both3.row_block(1)
        else:           # the three tables in one launch: one bracket search per wavelength for all 3 x nlayer rows
            stack = cld.__dict__.get("_stack")
            if stack is None:
                stack = cld.__dict__["_stack"] = np.concatenate([cld.compact[k] for k in ("opd", "w0", "g0")])
            all3 = regrid_rows(cld.in_wno, stack, d_x, ctx).reshape((3, nlayer, nwno))
            d_cld, d_w0, d_g0 = all3.row_block(0), all3.row_block(1), all3.row_block(2)
    else:
        d_cld = DeviceArray.from_host(taucld_host(), ctx)
        d_w0 = DeviceArray.from_host(plane(cld["w0"]), ctx)
        d_g0 = DeviceArray.from_host(plane(cld["g0"]), ctx)
    tm = 0
    if test_mode is not None:      # optics.py:372 `test_mode != None`: anything but None, including the
        tm = 1 if test_mode == "rayleigh" else 2            # signature default False, is a test mode
    out = {}
    for k in OUT_NAMES:
        rows = nlayer + 1 if k in ("tau", "tau_og") else nlayer
        out[k] = DeviceArray((rows,) + gshape, ctx) if (want is None or k in want) else None
    mix_args = (*[ptr(x.addr) if x is not None else None for x in (d_cld, d_w0, d_g0)],
                ptr(raman_plane.addr) if raman_plane else None, _ci(raman_rows),
                _cd(raman_const), _ci(tm), _ci(1 if delta_eddington else 0), _ci(stream),
                *[ptr(out[k].addr) if out[k] is not None else None for k in OUT_NAMES])
    if fused:
        gas_stage(atm, opa, None, None, mix=mix_args + (_ci(1),) + cld_tab)
        if tab_keep is not None:
            out["_cloud_tables"] = tab_keep          # the launch is asynchronous: its inputs live as long as its outputs
    else:
        check(load().picaso_compute_opacity_ck_dev(ctx, _ci(nlayer), _ci(nwno), _ci(ngauss), ptr(taugas.addr),
                                                   ptr(tauray.addr), *mix_args), ctx)
    out = {k: v for k, v in out.items() if v is not None}
    if full_output:
        def over_gauss(x):      # (nlayer, nwno) -> (nlayer, nwno, ngauss) as the reference stores it; a view for ngauss = 1
            return x[:, :, None] if ngauss == 1 else np.repeat(x[:, :, None], ngauss, axis=2)
        atmosphere.taugas = taugas.to_host().reshape((nlayer, nwno, ngauss))
        atmosphere.tauray = over_gauss(tauray.to_host())
        if getattr(atm, "cloud_free", False):                 # zeros, as the (read-only) cloud tables themselves
            atmosphere.taucld = np.broadcast_to(np.zeros((1, 1, 1)), (nlayer, nwno, ngauss))
        else:
            atmosphere.taucld = over_gauss(np.asarray(taucld_host()))
    return out


@_lib.serialized
def compute_opacity(atmosphere, opacityclass, ngauss=1, stream=2, delta_eddington=True,
                    test_mode=False, raman=0, plot_opacity=False, full_output=False,
                    return_mode=False, fthin_cld=None, do_holes=False):
    """Reference signature and return tuple (optics.py:26-27, 423-431): 13 numpy arrays
    ``(nlayer|nlevel, nwno, ngauss)``.  As in the reference, any ``test_mode`` other than ``None`` --
    including the signature default ``False`` -- selects a Dlugach test mode (optics.py:372);
    ``picaso()`` passes ``inputs['test_mode']``, which defaults to ``None``."""
    if plot_opacity or return_mode:
        raise Exception("plot_opacity / return_mode are plotting aids of the reference and are "
                        "not part of the accelerated path")
    d = compute_opacity_resident(atmosphere, opacityclass, ngauss=ngauss, stream=stream,
                                 delta_eddington=delta_eddington, test_mode=test_mode, raman=raman,
                                 fthin_cld=fthin_cld, do_holes=do_holes, full_output=full_output)
    return tuple(d[k].to_host().reshape(d[k].shape[:2] + (ngauss,)) for k in OUT_NAMES)


# ------------------------------------------------------------------------------------------------
# Raman factor (host side, once per atmosphere): reference optics.py:434-581
# ------------------------------------------------------------------------------------------------
def partition_function(j, T):
    """Statistical weight x Boltzmann factor of H2 rotational level J, restating reference
    optics.py:524-549 as written there (including its j(j+1) factor appearing in both ``b_energy``
    and the exponent, and the ortho/para weight 3 for odd J)."""
    k = 1.38064852e-16
    b = 60.853
    c = 29979245800
    h = 6.62607004e-27
    b_energy = (b * (h) * (c) * j * (j + 1) / k)
    if j % 2 == 0:
        return (2.0 * j + 1.0) * np.exp(-0.5 * b_energy * j * (j + 1) / T)
    return 3.0 * (2.0 * j + 1.0) * np.exp(-0.5 * b_energy * j * (j + 1) / T)


def partition_sum(T):
    """Partition sum truncated at J = 19 (reference optics.py:551-567)."""
    Z = np.zeros(np.size(T))
    for j in range(0, 20):
        Z += partition_function(j, T)
    return Z


def j_fraction(j, T):
    """Fraction of H2 in rotational state J at temperature T (reference optics.py:524-545)."""
    return partition_function(j, T) / partition_sum(T)


def compute_raman(nwno, nlayer, wno, stellar_shifts, tlayer, cross_sections, j_initial, deltanu):
    """Oklopcic Raman factor (reference optics.py:434-494)."""
    cross_sections = np.asarray(cross_sections, dtype=float)
    j_initial = np.asarray(j_initial, dtype=int)
    deltanu = np.asarray(deltanu, dtype=float)
    w_shift = np.zeros((nlayer, nwno))
    wo_shift = np.zeros((nlayer, nwno))
    ray = np.zeros((nlayer, nwno))
    j_at_temp = np.zeros((10, nlayer))
    for i in range(10):
        j_at_temp[i, :] = j_fraction(i, tlayer)
    for i in range(cross_sections.shape[0]):
        Q = cross_sections[i] / wno ** 3.0 / (wno + deltanu[i])
        if deltanu[i] == 0:
            ray += np.outer(j_at_temp[j_initial[i], :], Q)
        else:
            w_shift += np.outer(j_at_temp[j_initial[i], :], Q * stellar_shifts[:, i])
            wo_shift += np.outer(j_at_temp[j_initial[i], :], Q)
    return (ray + w_shift) / (ray + wo_shift)
