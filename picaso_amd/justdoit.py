"""API surface of the accelerated path: ``inputs`` / ``picaso()`` / ``inputs.spectrum()``
(counterpart of the reference ``picaso/justdoit.py``: ``picaso`` :65-621, ``class inputs`` :1421,
``phase_angle`` :1453, ``approx`` :4635, ``spectrum`` :4779).

What is kept: the call sequence, keyword surface and option tables for the 1-D reflected / thermal
spectrum with Toon two-stream radiative transfer, the routing of the 13 ``compute_opacity`` planes
into the solvers (reflected: delta-scaled + ``_OG``; thermal: ``DTAU_OG, W0_no_raman, COSB_OG``),
the return-dict keys and the post-processing formulas.  Quantities are plain cgs floats (the
reference's astropy-unit arguments are not reproduced) and everything between ``get_opacities`` and
``compress_disco`` stays in HBM.

Out of scope here: chemistry, virga clouds, stellar grids (``star()`` takes a
relative flux vector), xarray I/O, climate, retrievals, phase curves, 3-D regridding.
"""
import copy

import os

import numpy as np

from . import _lib, disco, optics, resident
from . import options as _options
from .device import DeviceArray
from .options import Options                                                        # noqa: F401  (jdi.Options)
from .spectrum import (Spectrum, setup_facets_3d, _atmosphere_block, _bond_denominator, _cloud_free_top, _constant_planes,   # noqa: F401
                       _fetch, _interp_axis, _ones, _post_final, _post_reflected, _post_thermal, _postprocess,
                       _reflected, _resident_vector, _setup_atmosphere, _trapz_resident)

# option tables (reference justdoit.py:5512-5534, 5647-5658)
def single_phase_options(printout=True):
    return ["cahoy", "OTHG", "TTHG", "TTHG_ray"]


def multi_phase_options(printout=True):
    return ["N=2", "N=1", "isotropic"]


def raman_options():
    return ["oklopcic", "pollack", "none"]


def toon_phase_coefficients(printout=True):
    return ["quadrature", "eddington"]


def SH_psingle_form_options(printout=True):
    return ["explicit", "legendre"]


def SH_scattering_options(printout=True):
    return ["TTHG", "OTHG", "isotropic"]


def SH_rayleigh_options(printout=True):
    return ["off", "on"]


_DEFAULTS = {   # reference/config.json
    "phase_angle": 0, "test_mode": None,
    "planet": {"gravity": None, "radius": np.nan, "mass": np.nan},
    "star": {"database": "nostar", "radius": "nostar", "semi_major": np.nan, "relative_flux": None},
    "atmosphere": {"profile": None, "exclude_mol": 1},
    "clouds": {"profile": None, "wavenumber": None, "do_holes": False},
    "approx": {"p_reference": 1, "rt_method": "toon", "get_lvl_flux": False,
               "rt_params": {"toon": {"toon_coefficients": 0, "multi_phase": 0, "single_phase": 3},
                             "SH": {"single_form": 0, "w_single_form": 0, "w_multi_form": 0,
                                    "psingle_form": 0, "w_single_rayleigh": 1, "w_multi_rayleigh": 1,
                                    "psingle_rayleigh": 1, "calculate_fluxes": 0},
                             "common": {"stream": 2, "delta_eddington": True, "raman": 1,
                                        "TTHG_params": {"fraction": [1, -1, 2], "constant_back": -0.5,
                                                        "constant_forward": 1}}}},
}


def _refdata():
    ref = os.environ.get("picaso_refdata")
    if ref is None:
        raise Exception("no file was given and the picaso_refdata environment variable is not set: the default "
                        "opacity files are looked up under $picaso_refdata/opacities, as in the reference")
    return ref


def opannection(wave_range=None, filename_db=None, resample=1, method="resampled", ck_db=None, raman_db=None,
                preload_gases="all", verbose=False, query_method="nearest", rayleigh_opa=None):
    """Opacity connection with the reference's keywords in the reference's order (``opannection``,
    justdoit.py:1296-1419): the tables are read once and stay resident in HBM.

    ``method='resampled'`` (default): monochromatic sqlite database ``filename_db`` (default, as in the
    reference, the first ``$picaso_refdata/opacities/opacities*.db``) -> ``RetrieveOpacities``;
    ``'preweighted'``: premixed correlated-k HDF5 file ``ck_db`` + continuum database ``filename_db`` (default
    ``$picaso_refdata/opacities/ck_cx_cont_opacities.db``) -> ``RetrieveCKs``; ``'resortrebin'``: directory
    ``ck_db`` of per-gas k-tables (``preload_gases``) mixed on the fly.  ``query_method`` ('nearest' |
    'linear': how (P,T) table points are combined, an attribute the reference sets on the class afterwards) and
    ``rayleigh_opa`` (Rayleigh cross sections as data) are additions after the reference's keywords."""
    import glob
    if isinstance(wave_range, (str, bytes, os.PathLike)):
        raise Exception("opannection(wave_range=None, filename_db=None, ...): the first positional argument is "
                        "wave_range, as in the reference; pass the database as filename_db=%r" % (wave_range,))
    if method == "resampled" and ck_db is None:
        if filename_db is None:
            found = sorted(glob.glob(os.path.join(_refdata(), "opacities", "opacities*.db")))
            if not found:
                raise Exception("Could not find anything with the naming scheme opacities*.db in the opacities "
                                "folder. Please use the get_data function to download an opacity file.")
            if len(found) > 1 and verbose:
                print("Found more than one opacity database. Choosing the first one.")
            filename_db = found[0]
        elif not os.path.isfile(filename_db):
            raise Exception("The opacity file you have entered does not exist: " + str(filename_db))
        if resample != 1 and verbose:
            print("YOU ARE REQUESTING RESAMPLING!! This could degrade the precision of your spectral calculations "
                  "so should be used with caution.")
        opa = optics.RetrieveOpacities.from_sqlite(filename_db, wave_range=wave_range, resample=resample,
                                                   query_method=query_method, rayleigh_opa=rayleigh_opa)
        # Raman cross sections of H2 (Oklopcic+2016): the reference reads `raman_db`, by default
        # $picaso_refdata/opacities/raman.txt, with every opacity object (optics.py:1956-1961).  Here a missing default
        # file is not an error until raman='oklopcic' asks for it.
        path = raman_db
        if path is None and os.environ.get("picaso_refdata") is not None:
            cand = os.path.join(_refdata(), "opacities", "raman.txt")
            path = cand if os.path.isfile(cand) else None
        if path is not None:
            opa.raman_db = read_raman_db(path)
        if verbose:
            print("verbose=True; Molecule set=", opa.molecules)
        return opa
    if method == "resampled":
        raise Exception("ck_db was supplied but method is set to resampled. Change kwarg method='preweighted' to "
                        "use the preweighted ck tables")
    if method == "preweighted":
        if filename_db is None:
            filename_db = os.path.join(_refdata(), "opacities", "ck_cx_cont_opacities.db")
        if ck_db is None or not os.path.exists(ck_db):
            if ck_db is not None and os.path.isfile(str(ck_db).rstrip("/") + ".tar.gz"):
                raise Exception("The CK filename that you have selected appears still be .tar.gz. Please unpack "
                                "and rerun")
            raise Exception("The CK filename that you have selected does not exist. Please make sure you have "
                            "downloaded and unpacked the right CK file.")
        return optics.RetrieveCKs.from_files(ck_db, filename_db, method="preweighted", rayleigh_opa=rayleigh_opa)
    if method == "resortrebin":
        if filename_db is None:
            filename_db = os.path.join(_refdata(), "climate_INPUTS", "ck_cx_cont_opacities_661.db")
        if ck_db is None:
            ck_db = os.path.join(_refdata(), "opacities", "resortrebin")
        return optics.RetrieveCKs.from_files(ck_db, filename_db, method="resortrebin", preload_gases=preload_gases,
                                             rayleigh_opa=rayleigh_opa)
    raise Exception("The only available opacity methods are: resortrebin, preweighted, and resampled")


_UNIT_CGS = {   # plain-string stand-ins for the astropy units of the reference's tutorials -> cgs factors
    "cm/s**2": 1.0, "cm/(s**2)": 1.0, "cm/s2": 1.0, "m/s**2": 100.0, "m/(s**2)": 100.0, "m/s2": 100.0,
    "cm": 1.0, "m": 100.0, "km": 1e5, "rjup": 7.1492e9, "r_jup": 7.1492e9, "jupiterrad": 7.1492e9,
    "rearth": 6.3781e8, "r_earth": 6.3781e8, "earthrad": 6.3781e8, "rsun": 6.957e10, "r_sun": 6.957e10,
    "g": 1.0, "kg": 1e3, "mjup": 1.8981246e30, "m_jup": 1.8981246e30, "jupitermass": 1.8981246e30,
    "mearth": 5.9721679e27, "m_earth": 5.9721679e27, "earthmass": 5.9721679e27, "msun": 1.98840987e33,
    "m_sun": 1.98840987e33, "au": 1.495978707e13,
}


def _to_cgs(value, unit, what):
    """``value * unit`` in cgs.  ``unit``: None (already cgs), a number (the cgs value of one unit), a unit
    string of ``_UNIT_CGS``, or an astropy unit / quantity when astropy is installed (duck-typed: ``.to``)."""
    if unit is None:
        return float(value)
    if isinstance(unit, (int, float, np.floating, np.integer)):
        return float(value) * float(unit)
    if isinstance(unit, str):
        key = unit.strip().lower().replace(" ", "")
        if key not in _UNIT_CGS:
            raise Exception("%s: unit %r not known; give a cgs value (unit=None), a factor, or one of %s"
                            % (what, unit, sorted(_UNIT_CGS)))
        return float(value) * _UNIT_CGS[key]
    if hasattr(unit, "to") or hasattr(value * unit, "to"):            # astropy
        target = {"gravity": "cm/s2", "radius": "cm", "mass": "g"}[what]
        return float((value * unit).to(target).value)
    raise Exception("%s: cannot interpret unit %r" % (what, unit))


_WAVE_UM = {"um": 1.0, "micron": 1.0, "microns": 1.0, "nm": 1e-3, "angs": 1e-4, "angstrom": 1e-4, "aa": 1e-4, "a": 1e-4,
            "cm": 1e4, "m": 1e6, "mm": 1e3}
# flux density per unit wavelength -> erg s^-1 cm^-2 per cm of wavelength (the reference's 'erg*cm^(-3)*s^(-1)')
_FLAM_CGS = {"erg/cm2/s/cm": 1.0, "erg/cm3/s": 1.0, "erg*cm^(-3)*s^(-1)": 1.0, "erg/s/cm3": 1.0,
             "flam": 1e8, "erg/cm2/s/angs": 1e8, "erg/cm2/s/aa": 1e8, "erg/cm2/s/a": 1e8, "erg/cm2/s/angstrom": 1e8,
             "erg/s/cm2/aa": 1e8, "erg/s/cm2/angs": 1e8, "erg/cm2/s/um": 1e4, "erg/s/cm2/um": 1e4, "erg/cm2/s/nm": 1e7,
             "w/m2/um": 1e7, "w/m2/nm": 1e10, "w/m2/m": 1e1, "w/m3": 1e1}


def _stellar_cgs(wave, flux, w_unit, f_unit):
    """A stellar spectrum per unit wavelength -> ``(wavenumber [cm^-1] increasing, flux [erg s^-1 cm^-3])``, what the
    reference takes from synphot (justdoit.py:1817-1824).  Unit strings of the tables above, or astropy units when
    astropy is installed."""
    wave, flux = np.asarray(wave, dtype=float), np.asarray(flux, dtype=float)
    if isinstance(w_unit, str) and isinstance(f_unit, str):
        wk, fk = w_unit.strip().lower().replace(" ", ""), f_unit.strip().lower().replace(" ", "")
        if wk not in _WAVE_UM:
            raise Exception("star: w_unit %r not known; one of %s" % (w_unit, sorted(_WAVE_UM)))
        if fk not in _FLAM_CGS:
            raise Exception("star: f_unit %r not known (flux per unit wavelength); one of %s" % (f_unit, sorted(_FLAM_CGS)))
        um, f = wave * _WAVE_UM[wk], flux * _FLAM_CGS[fk]
    else:
        try:
            import astropy.units as u
            um = (wave * u.Unit(w_unit)).to(u.um).value
            f = (flux * u.Unit(f_unit)).to(u.Unit("erg*cm^(-3)*s^(-1)"),
                                            equivalencies=u.spectral_density(um * u.um)).value
        except ImportError:
            raise Exception("star: give w_unit / f_unit as strings (astropy is not installed)")
    wno = 1e4 / um
    order = np.argsort(wno, kind="stable")
    return wno[order], f[order]


_PRESSURE_TO_BAR = {"bar": 1.0, "bars": 1.0, "pa": 1e-5, "pascal": 1e-5, "mbar": 1e-3, "millibar": 1e-3,
                    "dyn/cm2": 1e-6, "dyne/cm2": 1e-6, "barye": 1e-6, "atm": 1.01325, "hpa": 1e-3, "kpa": 1e-2}


def _dataset_like(ds, extra_coords=()):
    """``(coords, variables)`` of a GCM input in the reference's layout (justdoit.py:3414-3520, 4515-4600: an xarray
    Dataset with coordinates ``lon`` / ``lat`` (degrees) / ``pressure`` [/ ``wno``] and data variables on
    ``(lon, lat, pressure[, wno])``), taken from anything that looks like one -- an object with ``.coords[name].values``,
    ``.keys()`` and ``ds[name].values`` (xarray itself, when the caller has it), or a plain dictionary whose
    coordinate arrays sit under those names next to the variables (pressure unit: ``ds['pressure_unit']``, default
    bar).  xarray is not imported here.  Variables come back as ``(lon, lat, pressure[, wno])`` float arrays,
    ``coords['pressure']`` in bar."""
    names = ("lon", "lat", "pressure") + tuple(extra_coords)
    if hasattr(ds, "coords"):
        have = ds.coords
        for c in names:
            if c not in have:
                raise Exception('Must include "%s" as a coordinate. Please see GCM 3D input tutorials to learn how to '
                                'reformat your input.' % c)
        coords = {c: np.asarray(have[c].values, dtype=float) for c in names}
        unit = getattr(have["pressure"], "attrs", {}).get("units", "bar")
        variables = {}
        for k in ds.keys():
            v = ds[k]
            arr = np.asarray(v.values, dtype=float)
            dims = list(getattr(v, "dims", names[:arr.ndim]))
            order = [dims.index(c) for c in names if c in dims]
            variables[k] = np.transpose(arr, order) if order != list(range(arr.ndim)) else arr
    elif isinstance(ds, dict):
        for c in names:
            if c not in ds:
                raise Exception('Must include "%s" as a coordinate. Please see GCM 3D input tutorials to learn how to '
                                'reformat your input.' % c)
        coords = {c: np.asarray(ds[c], dtype=float) for c in names}
        unit = ds.get("pressure_unit", "bar")
        variables = {k: np.asarray(v, dtype=float) for k, v in ds.items()
                     if k not in names and k not in ("pressure_unit", "wno_unit")}
    else:
        raise Exception("PICASO has moved to only accept xarray input. Please see GCM 3D input tutorials to learn how "
                        "to reformat your input. (picaso_amd takes an xarray-like object or a dictionary with "
                        "lon / lat / pressure entries.)")
    fac = _PRESSURE_TO_BAR.get(str(unit).strip().lower())
    if fac is None:
        raise Exception("pressure unit %r not understood (bar, Pa, mbar, dyn/cm2, atm)" % (unit,))
    coords["pressure"] = coords["pressure"] * fac
    coords["pressure_unit_in"] = unit
    return coords, variables


def _regrid_lonlat(coords, variables, lon_new, lat_new):
    """Bilinear in (longitude, latitude), both in degrees, of every ``(lon, lat, ...)`` variable.  The reference
    hands this step to xesmf's 'bilinear' regridder (build_3d_input.py:12-62; not installable here, so the two are
    not compared): on a rectilinear lon / lat grid that is this interpolation up to ESMF's great-circle cell
    geometry."""
    out = {}
    lon = np.asarray(coords["lon"], dtype=float)
    # a longitude axis that goes around the planet is periodic: facets beyond its first / last column are interpolated
    # across the +-180 seam; a regional map (less than a full circle with its own spacing) keeps its end values
    ulon = np.unique(lon)
    wraps = ulon.size > 2 and (ulon[-1] - ulon[0]) + 1.5 * np.max(np.diff(ulon)) >= 360.0
    for k, v in variables.items():
        out[k] = _interp_axis(lat_new, coords["lat"], _interp_axis(lon_new, lon, v, 0, period=360.0 if wraps else None), 1)
    return out


def _interp_extrapolate(x, xp, fp):
    """Piecewise-linear through ``(xp, fp)``, the end segments continued outside (scipy ``interp1d(kind='linear',
    fill_value='extrapolate')``)."""
    j = np.clip(np.searchsorted(xp, x, side="right") - 1, 0, len(xp) - 2)
    slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j])
    return fp[j] + slope * (x - xp[j])


def bin_star(wno_new, wno_old, Fp):
    """Mean of the points of ``Fp`` (on the increasing grid ``wno_old``) in the top-hat bins around ``wno_new``
    (reference optics.py:497-521: half a grid step to either side, the lower edge closed except for the first bin);
    NaN where a bin catches none.  Searches instead of one mask per bin."""
    wno_new, wno_old, Fp = (np.asarray(a, dtype=float) for a in (wno_new, wno_old, Fp))
    n = wno_new.size
    delta = np.empty(n)
    delta[:-1] = wno_new[1:] - wno_new[:-1]
    delta[-1] = delta[-2]
    lo_edge = wno_new - 0.5 * np.concatenate(([delta[0]], delta[:-1]))
    hi_edge = wno_new + 0.5 * delta
    lo = np.searchsorted(wno_old, lo_edge, side="left")           # wno_old >= edge
    lo[0] = np.searchsorted(wno_old, lo_edge[0], side="right")    # first bin: wno_old > edge
    hi = np.searchsorted(wno_old, hi_edge, side="left")           # wno_old < edge
    # The bin means with the reference's own rounding: np.mean of the bin's points.  Bins are grouped by their
    # point count c and every group is one (nbins_c, c) gather reduced along its contiguous axis -- numpy sums each
    # row exactly as it sums the 1-D slice the reference hands to np.mean.  (A difference of running sums, as used
    # until round 3, loses relative precision in the faint bins of a steep spectrum; np.add.reduceat rounds
    # differently from np.mean from three points on.)
    cnt = hi - lo
    out = np.full(n, np.nan)
    for c in np.unique(cnt[cnt > 0]):
        sel = np.nonzero(cnt == c)[0]
        out[sel] = np.mean(Fp[lo[sel][:, None] + np.arange(c)[None, :]], axis=1)
    return out


def create_grid(min_wavelength, max_wavelength, constant_R):
    """Wavenumbers (increasing) of the constant-resolution wavelength grid from ``min_wavelength`` to at least
    ``max_wavelength`` micron: neighbours in the ratio (2R+1)/(2R-1) (reference opacity_factory.py:712-739)."""
    spacing = (2.0 * constant_R + 1.0) / (2.0 * constant_R - 1.0)
    npts = int(np.ceil(np.log(max_wavelength / min_wavelength) / np.log(spacing))) + 1
    wl = np.cumprod(np.concatenate(([min_wavelength], np.full(npts - 1, spacing))))
    return 1e4 / wl[::-1]


def mean_regrid(x, y, newx=None, R=None):
    """Bin ``y(x)`` to the grid ``newx`` (bin edges half way between its points) or to constant resolution ``R``: the
    mean of the points in every bin, NaN where there is none (reference justplotit.py:31-63, which uses
    ``scipy.stats.binned_statistic``; the same counting here with numpy).  Returns ``(bin centres, means)``."""
    x, y = np.asarray(x, dtype=float), np.asarray(y, dtype=float)
    if newx is None and R is not None:
        edges = create_grid(1e4 / np.max(x), 1e4 / np.min(x), R)
    elif newx is not None and R is None:
        newx = np.asarray(newx, dtype=float)
        d = np.diff(newx)
        edges = np.concatenate(([newx[0] - d[0] / 2], newx[:-1] + d / 2.0, [newx[-1] + d[-1] / 2]))
    else:
        raise Exception("Please either enter a newx or a R")
    nb = edges.size - 1
    idx = np.searchsorted(edges, x, side="right") - 1
    idx[x == edges[-1]] = nb - 1                                  # the last bin includes its right edge
    ok = (idx >= 0) & (idx < nb)
    sums = np.bincount(idx[ok], weights=y[ok], minlength=nb)
    cnt = np.bincount(idx[ok], minlength=nb)
    with np.errstate(invalid="ignore", divide="ignore"):
        means = np.where(cnt > 0, sums / cnt, np.nan)
    return (edges[:-1] + edges[1:]) / 2.0, means


def read_raman_db(path):
    """The reference's ``raman.txt`` (Oklopcic, Hirata & Heng 2016, table of H2 Raman cross sections): 16 header lines,
    then whitespace columns ji, jf, vf, c, deltanu (reference optics.py:1956-1961) -> ``{'ji', 'jf', 'vf', 'c',
    'deltanu'}`` arrays, what ``compute_raman`` and ``star()`` take."""
    if not os.path.isfile(path):
        raise Exception("raman_db %s not found" % path)
    try:                                # pandas' float parser, as the reference: its last bit can differ from strtod's
        import pandas as pd
        tab = pd.read_csv(path, sep=r"\s+", skiprows=16, header=None, names=["ji", "jf", "vf", "c", "deltanu"]).values
    except ImportError:
        tab = np.loadtxt(path, skiprows=16, ndmin=2)
    if tab.ndim != 2 or tab.shape[1] < 5:
        raise Exception("raman_db %s: expected the columns ji, jf, vf, c, deltanu after 16 header lines" % path)
    return {"ji": tab[:, 0].astype(int), "jf": tab[:, 1].astype(int), "vf": tab[:, 2].astype(int), "c": tab[:, 3].copy(),
            "deltanu": tab[:, 4].copy()}


class _UnitNames:
    """``jdi.u.Unit('m/(s**2)')`` of the reference's tutorials without astropy: the name itself, which ``gravity``,
    ``star`` and the other builders take (``_UNIT_CGS``)."""

    @staticmethod
    def Unit(name):
        return name


try:                                    # the reference exposes astropy.units as `justdoit.u`
    import astropy.units as u           # noqa: F401
except ImportError:
    u = _UnitNames()


def _base_case(name):
    return os.path.join(_refdata(), "base_cases", name)


def jupiter_pt():
    """Path of the reference's Jupiter P-T-composition profile (``$picaso_refdata/base_cases``, justdoit.py:5415)."""
    return _base_case("jupiter.pt")


def jupiter_cld():
    return _base_case("jupiterf3.cld")


def HJ_pt():
    return _base_case("HJ.pt")


def HJ_cld():
    return _base_case("HJ.cld")


def brown_dwarf_pt():
    return _base_case("t1270g200f1_m0.0_co1.0.cmp")


def brown_dwarf_cld():
    return _base_case("t1270g200f1_m0.0_co1.0.cld")


def get_cld_input_grid(filename_or_grid="wave_EGP.dat", grid661=False):
    """Wavenumbers (increasing) of the 196-point grid cloud tables come on (reference wavelength.py:9-40):
    ``$picaso_refdata/opacities/wave_EGP.dat`` (whitespace table with a 'wavenumber' column) or an array;
    ``grid661``: the 661-point grid of the climate tables, ``$picaso_refdata/climate_INPUTS/wvno_661``."""
    if grid661:
        return np.loadtxt(os.path.join(_refdata(), "climate_INPUTS", "wvno_661"), usecols=[0, 1], unpack=True)[0]
    if isinstance(filename_or_grid, np.ndarray):
        return np.sort(filename_or_grid)
    path = filename_or_grid
    if not os.path.isabs(path):
        path = os.path.join(_refdata(), "opacities", filename_or_grid)
    if not os.path.isfile(path):
        raise Exception("cloud wavenumber grid %s not found" % path)
    try:                                # the reference's reader (pandas' float parser), wavelength.py:31-32
        import pandas as pd
        grid = pd.read_csv(path, sep=r"\s+")
        if "wavenumber" not in grid.keys():
            raise Exception('Please make sure there is a column named "wavenumber" in your cloud wavegrid file')
        return np.sort(np.asarray(grid["wavenumber"].values, dtype=float))
    except ImportError:
        pass
    with open(path) as fh:
        header = fh.readline().split()
        col = header.index("wavenumber")
        vals = [float(line.split()[col]) for line in fh if line.strip()]
    return np.sort(np.array(vals))


class inputs:
    """Builder of the run configuration (reference ``class inputs``, justdoit.py:1421)."""

    def __init__(self, calculation="planet", climate=False):
        """``calculation='browndwarf'``: no star and no Raman scattering (``setup_nostar``), as in the reference
        (justdoit.py:1446-1451).  ``climate=True`` would start the T(P) iteration, which is outside this package
        (its radiative-transfer call is ``picaso_amd.climate.get_fluxes``)."""
        if climate:
            raise Exception("inputs(climate=True): the climate solver's T(P) iteration is outside this package; its "
                            "radiative-transfer call is picaso_amd.climate.calculate_atm / get_fluxes")
        self.inputs = copy.deepcopy(_DEFAULTS)
        self.inputs["calculation"] = calculation
        if "brown" in calculation:
            self.setup_nostar()
        self.phase_angle(0)

    def setup_nostar(self):
        """Turns off the planet-specific things (reference justdoit.py:1740-1754): no star, no Raman scattering."""
        self.inputs["approx"]["rt_params"]["common"]["raman"] = 2
        self.inputs["star"].update(database="nostar", temp="nostar", logg="nostar", metal="nostar", radius="nostar",
                                   radius_unit="nostar", flux="nostar", wno="nostar", relative_flux=None)

    def clouds_reset(self):
        """Cloud tables back to zeros (reference justdoit.py:4115-4124)."""
        prof = self.inputs["clouds"]["profile"]
        if prof is not None:
            self.inputs["clouds"]["profile"] = {k: (np.zeros_like(np.asarray(v, dtype=float)) if k in ("g0", "w0", "opd") else v)
                                                for k, v in dict(prof).items()}

    def phase_angle(self, phase=0, num_gangle=10, num_tangle=1, symmetry=False, phase_grid=None, calculation=None):
        """Geometry (reference justdoit.py:1453-1605), including the quadrant ``symmetry`` reduction; with
        ``phase_grid`` the geometry of every phase of a phase curve (``phase_curve_geometry``, :1492-1497)."""
        if phase_grid is not None:
            if calculation is None:
                raise Exception("Phase curve calculation activated because phase_grid is supplied. However, 'calculation' "
                                "needs to be specified to either 'thermal' or 'reflected'")
            self.phase_curve_geometry(calculation, phase_grid, num_gangle=num_gangle, num_tangle=num_tangle)
            return
        if (phase > 2 * np.pi) or (phase < 0):
            raise Exception("Oops! you input a phase angle greater than 2*pi or less than 0. Please "
                            "make sure your inputs are in radian units: 0<phase<2pi")
        if (num_tangle == 1) or (num_gangle == 1):
            if phase != 0:
                raise Exception("The default PICASO disk integration is to use num_tangle=1 and "
                                "num_gangle>1 ... please resubmit phase_angle with num_tange>10 and "
                                "num_gangle>10.")
            if num_gangle == 1:
                raise Exception("num_gangle cannot be 1. Please resubmit your run with num_tangle=1, "
                                "and increase number of Gauss points")
            num_gangle = int(num_gangle / 2)
            possible = np.array([5, 6, 7, 8])
            num_gangle = int(possible[(np.abs(possible - num_gangle)).argmin()])
            gangle, gweight, tangle, tweight = disco.get_angles_1d(num_gangle)
            ng, nt = len(gangle), len(tangle)
            ubar0, ubar1, cos_theta, lat, lon = disco.compute_disco(ng, nt, gangle, tangle, phase)
            cos_theta = 1.0                                   # justdoit.py:1532
        else:
            ng, nt = int(num_gangle), int(num_tangle)
            gangle, gweight, tangle, tweight = disco.get_angles_3d(ng, nt)
            ubar0, ubar1, cos_theta, lat, lon = disco.compute_disco(ng, nt, gangle, tangle, phase)
            if symmetry:                                      # justdoit.py:1562-1602: one quadrant
                if phase != 0:
                    raise Exception("If phase is non zero then you cannot utilize symmetry to reduce "
                                    "computation speed.")
                for n, nm in ((num_tangle, "num_tangle"), (num_gangle, "num_gangle")):
                    if n == 2 or (n % 2 != 0 and n != 1):
                        raise Exception("Youve selected %s=%d however for symmetry to be utilized we need "
                                        "at LEAST two points on either side of the symmetric axis (e.g. "
                                        "num angles >=4 or 1)." % (nm, n))
                full = dict(num_gangle=ng, num_tangle=nt, gangle=gangle, gweight=gweight, tangle=tangle,
                            tweight=tweight, latitude=lat, longitude=lon, cos_theta=cos_theta, ubar0=ubar0,
                            ubar1=ubar1)
                nt_uni = len(np.unique((tweight * 1e6).astype(int)))          # unique to 1 ppm
                ng_uni = len(np.unique((gweight * 1e6).astype(int)))
                self.inputs["phase_angle"] = phase
                self.inputs["disco"] = dict(
                    symmetry="true", num_tangle=nt_uni, num_gangle=ng_uni, ubar1=ubar1[0:ng_uni, 0:nt_uni],
                    ubar0=ubar0[0:ng_uni, 0:nt_uni], latitude=lat[0:nt_uni], longitude=lon[0:ng_uni],
                    gangle=gangle[0:ng_uni], gweight=(num_tangle / nt_uni) * gweight[0:ng_uni],   # sic (:1595-1598)
                    tangle=tangle[0:nt_uni], tweight=(num_gangle / ng_uni) * tweight[0:nt_uni],
                    cos_theta=cos_theta, full_geometry=full)
                return
        self.inputs["phase_angle"] = phase
        self.inputs["disco"] = dict(num_gangle=ng, num_tangle=nt, gangle=gangle, gweight=gweight,
                                    tangle=tangle, tweight=tweight, latitude=lat, longitude=lon,
                                    cos_theta=cos_theta, ubar0=ubar0, ubar1=ubar1)
        if nt > 1:
            self.inputs["disco"]["symmetry"] = "false"

    def gravity(self, gravity=None, gravity_unit=None, radius=None, radius_unit=None, mass=None, mass_unit=None):
        """Surface gravity from ``gravity`` or from ``radius`` and ``mass``, with the reference's keywords in the
        reference's order (justdoit.py:1663-1702).  A ``*_unit`` of ``None`` means the value is already cgs
        (cm/s^2, cm, g); otherwise a unit string ('m/s**2', 'rjup', 'mjup', 'rearth', ...), a cgs factor, or an
        astropy unit when astropy is installed."""
        if radius is not None and isinstance(radius, float) and np.isnan(radius):
            radius = None
        if mass is not None and isinstance(mass, float) and np.isnan(mass):
            mass = None
        planet = self.inputs["planet"]
        if (mass is not None) and (radius is not None):
            m, r = _to_cgs(mass, mass_unit, "mass"), _to_cgs(radius, radius_unit, "radius")
            g = 6.6743e-8 * m / r ** 2              # as the reference: mass and radius win over `gravity`
            planet.update(radius=r, radius_unit="cm", mass=m, mass_unit="g", gravity=g, gravity_unit="cm/(s**2)")
        elif gravity is not None:
            planet.update(gravity=_to_cgs(gravity, gravity_unit, "gravity"), gravity_unit="cm/(s**2)",
                          radius=np.nan, radius_unit="Radius not specified", mass=np.nan,
                          mass_unit="Mass not specified")
        else:
            raise Exception("Need to specify gravity or radius and mass + additional units")

    def star(self, opannection=None, temp=None, metal=None, logg=None, radius=None, radius_unit=None, semi_major=None,
             semi_major_unit=None, database="ck04models", filename=None, w_unit=None, f_unit=None, relative_flux=None):
        """The star, with the reference's keywords in the reference's order (justdoit.py:1756-1906).

        ``filename`` (two whitespace columns: wavelength, flux; ``w_unit`` / ``f_unit`` name their units) is binned onto
        the opacity grid of ``opannection`` the way the reference does it: mean of the stellar points in every bin with
        interpolated values where a bin catches none (:1880-1888); with ``approx(raman='oklopcic')`` the shifted /
        unshifted ratios of ``compute_stellar_shits`` on a 5x finer grid (:1833-1842); with ``get_lvl_flux`` the
        bin-integrated form (:1843-1879).  ``relative_flux`` (an addition): the flux already on the opacity grid
        (times ``(radius/semi_major)**2`` when both are given it is what the solvers take as F0PI).  Stellar model
        grids (``temp, metal, logg`` through stsynphot) are outside this package: the call says so.  Nothing given:
        no star (F0PI = 1, justdoit.py:174-177).  Units: ``None`` = cgs, a unit string (``'R_sun'``, ``'au'``, ...), a cgs
        factor, or an astropy unit when astropy is installed."""
        st = self.inputs["star"]
        r = _to_cgs(radius, radius_unit, "radius") if radius is not None else np.nan
        sm = _to_cgs(semi_major, semi_major_unit, "radius") if semi_major is not None else np.nan
        if relative_flux is not None:
            st.update(database="user", radius=r, semi_major=sm, relative_flux=np.asarray(relative_flux, dtype=float))
            return
        if filename is None:
            if temp is not None or metal is not None or logg is not None:
                raise Exception("star(temp=, metal=, logg=) looks a model up in the stsynphot grids, which are outside this "
                                "package: give the spectrum as filename= (with w_unit, f_unit) or relative_flux=")
            st.update(database="nostar", radius="nostar", relative_flux=None)
            return
        if opannection is None:
            raise Exception("star(filename=...) needs the opacity object: its wavenumber grid is what the star is binned to")
        if w_unit is None or f_unit is None:
            raise Exception("Must enter 1) filename,w_unit & f_unit OR 2)temp, metal & logg ")
        tab = np.loadtxt(filename, usecols=(0, 1), ndmin=2)
        wno_star, flux_star = _stellar_cgs(tab[:, 0], tab[:, 1], w_unit, f_unit)
        wno = np.asarray(opannection.wno, dtype=float)
        lvl = bool(self.inputs["approx"].get("get_lvl_flux", False))
        if self.inputs["approx"]["rt_params"]["common"]["raman"] == 0:           # :1833-1842
            if getattr(opannection, "raman_db", None) is None:
                raise Exception("raman='oklopcic' needs the Raman cross-section table: opannection(raman_db=...)")
            fine = np.linspace(wno.min() - 2000.0, wno.max() + 6000.0, wno.size * 5)
            fine_flux = np.interp(fine, wno_star, flux_star)
            dnu = np.asarray(opannection.raman_db["deltanu"], dtype=float)
            shifts = np.zeros((wno.size, dnu.size))
            opannection.unshifted_stellar_spec = bin_star(wno, fine, fine_flux)
            first = None
            for i in range(dnu.size):                                             # optics.py:2394-2400
                sh = bin_star(wno + dnu[i], fine, fine_flux)
                if first is None:
                    first = sh * 0 + sh
                shifts[:, i] = sh / first
            opannection.raman_stellar_shifts = shifts
            binned, unit = opannection.unshifted_stellar_spec, "ergs cm^{-2} s^{-1} cm^{-1}"
        elif lvl:                                                                 # :1843-1879
            if np.isnan(r) or np.isnan(sm):
                raise Exception("semi_major and r parameters are not provided but are needed to compute relative fluxes "
                                "for climate calculation or when get_lvl_flux is True")
            ok = flux_star > 1e-30
            lw, lf = np.log10(wno_star[ok]), np.log10(flux_star[ok])
            fine = 10 ** _interp_extrapolate(np.log10(wno), lw, lf)
            x = -1.0 / wno
            binned = np.zeros(wno.size)
            binned[:-1] = np.diff(x) * (fine[1:] + fine[:-1]) / 2.0               # the two grid points of a bin, :1858-1862
            if wno.size > 2:
                slope = (binned[-2] - binned[-3]) / (wno[-2] - wno[-3])
                binned[-1] = binned[-2] + slope * (wno[-1] - wno[-2])
            bad = np.isnan(binned) | (binned == 0)
            if bad.sum() > 20:
                good = np.where(~bad)[0]
                binned[bad] = np.interp(wno[bad], wno[good], binned[good])
            opannection.unshifted_stellar_spec = binned
            unit = "ergs cm^{-2} s^{-1}"
        else:                                                                     # :1880-1888
            interp = np.interp(wno, wno_star, flux_star)
            _, binned = mean_regrid(wno_star, flux_star, newx=wno)
            empty = np.isnan(binned)
            binned[empty] = interp[empty]
            opannection.unshifted_stellar_spec = binned
            unit = "ergs cm^{-2} s^{-1} cm^{-1}"
        have = not (np.isnan(r) or np.isnan(sm))
        opannection.relative_flux = binned * (r / sm) ** 2 if have else binned * 0 + 1
        st.update(database=database, temp=temp, logg=logg, metal=metal, radius=r, radius_unit="cm" if radius is not None
                  else "Radius not supplied", flux=binned, flux_unit=unit, relative_flux=opannection.relative_flux,
                  relative_flux_unit="(Rs/Sa)^2 * " + unit, semi_major=sm,
                  semi_major_unit="cm" if semi_major is not None else "Semi Major axis not supplied", filename=filename,
                  w_unit=w_unit, f_unit=f_unit)

    def atmosphere(self, df=None, filename=None, exclude_mol=None, mh=None, cto_absolute=None, cto_relative=None,
                   chem_method=None, quench=False, no_ph3=False, cold_trap=False, vol_rainout=False,
                   photochem_init_args=None, add_visscher_abunds=True, **pd_kwargs):
        """Level profile with the reference's keywords (justdoit.py:1915-2075): ``df`` (DataFrame or dict) or
        ``filename`` (read with ``pandas.read_csv(filename, **pd_kwargs)``) with columns pressure [bar], temperature [K]
        and volume mixing ratios.  As in the reference: levels are sorted by pressure; ``exclude_mol`` (a name or a list
        of names) switches molecules off in the opacities only; and Raman scattering is switched off
        (``raman='none'``) for an atmosphere without H2 or with less than 70 % of it anywhere (:2033-2040).
        The chemistry keywords (``mh``, ``cto_*``, ``chem_method``, photochemistry, the climate hacks) belong to
        subsystems outside this package and raise."""
        if any(x is not None for x in (mh, cto_absolute, cto_relative, chem_method, photochem_init_args)):
            raise Exception("atmosphere(mh=, cto_*=, chem_method=, photochem_init_args=): chemistry is outside this package; "
                            "give the mixing ratios as columns of df / filename")
        if any((quench, no_ph3, cold_trap, vol_rainout)):
            raise Exception("'quench','no_ph3','cold_trap','vol_rainout' are a climate kwargs and climate calculation is "
                            "not specified so this will not do anything")
        if df is not None:
            if not (isinstance(df, dict) or (hasattr(df, "keys") and hasattr(df, "sort_values"))):
                raise Exception("df must be pandas DataFrame or dictionary")
        elif filename is not None:
            import pandas as pd
            df = pd.read_csv(filename, **pd_kwargs)
        elif self.inputs["atmosphere"]["profile"] is not None:
            df = self.inputs["atmosphere"]["profile"]
        else:
            raise Exception("Could not find a starting dataframe in inputs['atmosphere']['profile'] and no df or filename "
                            "were specified")
        if "pressure" not in df.keys():
            raise Exception("Check column names. `pressure` must be included.")
        if "temperature" not in df.keys():
            raise Exception("`temperature` not specified as a column/key name")
        if exclude_mol is None or (isinstance(exclude_mol, (int, float)) and exclude_mol == 1):
            self.inputs["atmosphere"]["exclude_mol"] = 1
        elif isinstance(exclude_mol, dict):
            self.inputs["atmosphere"]["exclude_mol"] = exclude_mol
        else:
            names = [exclude_mol] if isinstance(exclude_mol, str) else list(exclude_mol)
            flags = {k: 1 for k in df.keys()}
            flags.update({m: 0 for m in names})
            self.inputs["atmosphere"]["exclude_mol"] = flags
        if hasattr(df, "sort_values"):
            df = df.sort_values("pressure").reset_index(drop=True)
        else:
            pres = np.asarray(df["pressure"], dtype=float)
            if np.any(np.diff(pres) < 0):
                order = np.argsort(pres, kind="stable")
                df = {k: np.asarray(v)[order] for k, v in df.items()}
        self.inputs["atmosphere"]["profile"] = df
        self.nlevel = len(df["pressure"])
        common = self.inputs["approx"]["rt_params"]["common"]
        if len(df.keys()) > 2 and common["raman"] != 2:
            if "H2" not in df.keys() or float(np.min(np.asarray(df["H2"], dtype=float))) < 0.7:
                common["raman"] = 2

    def clouds(self, filename=None, g0=None, w0=None, opd=None, p=None, dp=None, df=None, do_holes=False,
               fhole=None, fthin_cld=None, wavenumber=None, **pd_kwargs):
        """Cloud ``opd`` / ``w0`` / ``g0`` per (layer, wavelength) with the reference's keywords in the reference's
        order (justdoit.py:4126-4268): a table (``df`` dict / DataFrame, or ``filename`` read with pandas), or box
        clouds -- lists ``g0, w0, opd`` with bottom pressures ``p`` and thicknesses ``dp`` (log10 bar), laid on the
        196-point cloud wavenumber grid of ``$picaso_refdata/opacities/wave_EGP.dat`` exactly as the reference does.
        ``wavenumber`` (an addition): the table's own grid when it has no 'wavenumber' column and is not on the
        opacity grid."""
        if not hasattr(self, "nlevel"):
            raise Exception("Please make sure to run `atmosphere` before adding clouds")
        nlayer = self.nlevel - 1
        if (filename is not None) != (df is not None):
            if filename is not None:
                import pandas as pd
                df = pd.read_csv(filename, **pd_kwargs)
            cols = list(df.keys())
            for k in ("g0", "w0", "opd"):
                if k not in cols:
                    raise Exception("Please make sure %s is a named column in cld file" % k)
            if ("pressure" in cols) and ("wavenumber" in cols):      # sort by (pressure, wavenumber), :4211-4216
                pr, wn = np.asarray(df["pressure"], dtype=float), np.asarray(df["wavenumber"], dtype=float)
                order = np.lexsort((wn, pr))
                grid = np.unique(wn)
                if pr.size != nlayer * grid.size:
                    raise Exception("There are %d rows in the df, which does not equal %d layers previously "
                                    "specified x %d wave pts" % (pr.size, nlayer, grid.size))
                df = {k: np.asarray(df[k], dtype=float)[order] for k in ("opd", "w0", "g0")}
                wavenumber = grid
            elif wavenumber is None:
                # a table without its own grid (eddysed / virga output): the 196-point grid of wave_EGP.dat, or the
                # 661-point grid of the climate tables, told by its length (justdoit.py:4212-4219); this package also
                # takes tables that are already on the opacity grid (told apart in get_clouds)
                rows = np.size(df["opd"])
                if rows == nlayer * 196 and os.environ.get("picaso_refdata") is not None \
                        and os.path.isfile(os.path.join(_refdata(), "opacities", "wave_EGP.dat")):
                    wavenumber = get_cld_input_grid("wave_EGP.dat")
                elif rows == nlayer * 661 and os.environ.get("picaso_refdata") is not None \
                        and os.path.isfile(os.path.join(_refdata(), "climate_INPUTS", "wvno_661")):
                    wavenumber = get_cld_input_grid(grid661=True)
            self.inputs["clouds"].update(profile=df, wavenumber=wavenumber)
        elif filename is not None:
            raise Exception("give either filename or df, not both")
        elif None in [g0, w0, opd, p, dp]:
            raise Exception("Must either give dataframe/dict, OR a complete set of g0, w0, opd,p,dp to compute cloud "
                            "profile")
        else:                                                       # box clouds, justdoit.py:4235-4266
            plev = np.asarray(self.inputs["atmosphere"]["profile"]["pressure"], dtype=float)
            player = np.sqrt(plev[1:] * plev[:-1])
            wgrid = get_cld_input_grid("wave_EGP.dat")
            prof = {k: np.zeros((nlayer, wgrid.size)) for k in ("g0", "w0", "opd")}
            for ig, iw, io, ip, idp in zip(*[np.atleast_1d(x) for x in (g0, w0, opd, p, dp)]):
                inside = (player >= 10 ** (ip - idp)) & (player <= 10 ** ip)
                prof["g0"][inside], prof["w0"][inside], prof["opd"][inside] = ig, iw, io
            self.inputs["clouds"].update(profile=prof, wavenumber=wgrid)
        self.inputs["clouds"]["do_holes"] = do_holes
        if do_holes:
            if fhole is None:
                raise Exception("fhole must be float 0-1 if do_holes = True")
            self.inputs["clouds"].update(fhole=fhole, fthin_cld=fthin_cld)

    def atmosphere_3d(self, ds, regrid=True, plot=True, iz_plot=0, verbose=True, exclude_mol=1):
        """The reference's call (justdoit.py:3414-3519): ``ds`` = the GCM dataset, temperature and abundances on
        ``(lon, lat, pressure)`` with coordinates in degrees and a pressure unit (an xarray Dataset where the caller
        has xarray, or a dictionary, see ``_dataset_like``).  ``regrid=True`` interpolates it bilinearly onto the
        Gauss / Chebyshev facets of ``phase_angle()``; ``regrid=False`` checks that its grid IS that one, as the
        reference does.  ``plot`` / ``iz_plot`` are accepted and ignored (no plotting here).

        Facet profiles that are already arrays are taken as before: a dictionary WITHOUT ``lon`` / ``lat`` --
        ``pressure`` (nlevel,) in bar, ``temperature`` and every mixing ratio ``(nlevel, num_gangle, num_tangle)``
        or ``(nlevel,)`` (shared by all facets)."""
        if isinstance(ds, dict) and not hasattr(ds, "coords") and "lon" not in ds and "lat" not in ds:
            profiles = ds
            if "pressure" not in profiles or "temperature" not in profiles:
                raise Exception("atmosphere_3d(profiles) needs 'pressure' and 'temperature'")
            self.inputs["atmosphere"]["profile_3d"] = {k: np.asarray(v, dtype=float) for k, v in profiles.items()}
            self.inputs["atmosphere"]["exclude_mol"] = exclude_mol
            self.nlevel = len(profiles["pressure"])
            return
        coords, variables = _dataset_like(ds)
        if "temperature" not in variables:
            raise Exception("Must include temperature as data component")
        if verbose and str(coords["pressure_unit_in"]).lower() not in ("bar", "bars"):
            print("verbose=True; Converting pressure grid from %s to required unit of bar." % coords["pressure_unit_in"])
        geom = self.inputs["disco"]
        ng, nt, phase = geom["num_gangle"], geom["num_tangle"], self.inputs["phase_angle"]
        lat_f, lon_f = geom["latitude"] * 180 / np.pi, geom["longitude"] * 180 / np.pi
        if regrid:
            assert nt <= len(coords["lat"]), \
                "Cannot regrid from a course grid. num_tangle=%d and input grid has len(lat)=%d" % (nt, len(coords["lat"]))
            assert ng <= len(coords["lon"]), \
                "Cannot regrid from a course grid. num_gangle=%d and input grid has len(lon)=%d" % (ng, len(coords["lon"]))
            if verbose:
                print("verbose=True;regrid=True; Regridding 3D output to ngangle=%d, ntangle=%d, with phase=%s."
                      % (ng, nt, phase))
            variables = _regrid_lonlat(coords, variables, lon_f, lat_f)
        else:
            assert np.array_equal(lat_f, coords["lat"]), \
                "Latitudes from the GCM do not match the PICASO grid (phase %s): provide the native GCM grid with regrid=True" % phase
            assert np.array_equal(lon_f, coords["lon"]), \
                "Longitude from the GCM do not match the PICASO grid (phase %s): provide the native GCM grid with regrid=True" % phase
        if len(variables) == 1 and verbose:
            print("verbose=True;Only one data variable included. Make sure to add in chemical abundances before "
                  "trying to run spectra.")
        order = np.argsort(coords["pressure"])                         # ds.sortby('pressure')
        prof = {"pressure": coords["pressure"][order]}
        for k, v in variables.items():                                 # (lon, lat, pressure) -> (nlevel, ng, nt)
            prof[k] = np.ascontiguousarray(np.transpose(v[:, :, order], (2, 0, 1)))
        self.inputs["atmosphere"]["profile_3d"] = prof
        self.inputs["atmosphere"]["exclude_mol"] = exclude_mol
        self.nlevel = len(prof["pressure"])

    def phase_curve_geometry(self, calculation, phase_grid, num_gangle=10, num_tangle=10):
        """Facet geometry of every phase of a phase curve (reference justdoit.py:1606-1660): reflected
        light takes the illumination geometry of each phase, thermal emission the full-disk geometry
        of phase 0 for all of them (the planet emits in every direction; the phase enters through
        the rotated temperature map)."""
        phase_grid = [float(p) for p in phase_grid]
        if min(phase_grid) < 0:
            raise Exception("Input minimum of phase grid less than 0. Input phase_grid such that there are "
                            "only values between 0-2pi")
        if max(phase_grid) > np.pi * 2:
            raise Exception("Input maximum of phase grid is greater than 2pi. Input phase_grid such that "
                            "there are only values between 0-2pi")
        if calculation not in ("thermal", "reflected"):
            raise Exception("Phase curve setup only works for calculation=thermal or reflected")
        ng, nt = int(num_gangle), int(num_tangle)
        gangle, gweight, tangle, tweight = disco.get_angles_3d(ng, nt)

        def compute_angles(phase):
            ubar0, ubar1, cos_theta, lat, lon = disco.compute_disco(ng, nt, gangle, tangle, phase)
            return dict(num_gangle=ng, num_tangle=nt, gangle=gangle, gweight=gweight, tangle=tangle,
                        tweight=tweight, latitude=lat, longitude=lon, cos_theta=cos_theta, ubar0=ubar0,
                        ubar1=ubar1, symmetry="false")
        self.inputs["phase_angle"] = phase_grid
        self.inputs["disco"] = {p: compute_angles(0.0 if calculation == "thermal" else p) for p in phase_grid}
        self.inputs["disco"]["calculation"] = calculation

    def atmosphere_4d(self, ds=None, shift=None, plot=True, iz_plot=0, verbose=True, zero_point="night_transit",
                      exclude_mol=1):
        """The profiles of every phase of ``phase_curve_geometry`` (reference justdoit.py:3666-3880).

        ``ds`` = a list with one ``atmosphere_3d``-style facet-profile dictionary per phase (arrays already on the
        facets of each phase), or the reference's form: ONE GCM dataset (see ``atmosphere_3d``), which is rotated by
        ``phase + shift`` degrees of longitude per phase (``shift``: one value per phase, default 0;
        ``zero_point='night_transit'`` adds 180 degrees for thermal curves as the reference does) and interpolated
        bilinearly onto that phase's facets.  The reference additionally re-centres the illuminated crescent of a
        reflected-light curve with corrections written for its 6- and 10-angle grids (justdoit.py:3765-3830);
        those are not reproduced: hand per-phase facet profiles over for such a curve."""
        phases = self.inputs["phase_angle"]
        if ds is None:
            raise Exception("Need to submit the GCM dataset (or one facet-profile dictionary per phase)")
        if isinstance(ds, (list, tuple)):
            profs = list(ds)
            for pr in profs:
                if "pressure" not in pr or "temperature" not in pr:
                    raise Exception("atmosphere_4d: every phase needs 'pressure' and 'temperature'")
            self.inputs["atmosphere"]["profile_4d"] = [{k: np.asarray(v, dtype=float) for k, v in pr.items()}
                                                       for pr in profs]
            self.inputs["atmosphere"]["exclude_mol"] = exclude_mol
            self.nlevel = len(profs[0]["pressure"])
            return
        all_geom = self.inputs["disco"]
        if not isinstance(all_geom, dict) or "calculation" not in all_geom:
            raise Exception("run phase_curve_geometry() first")
        shift = np.zeros(len(phases)) if shift is None else np.asarray(shift, dtype=float)
        if len(shift) != len(phases):
            raise Exception("shift must have one entry per phase (%d)" % len(phases))
        if zero_point == "night_transit":
            if "reflected" in all_geom["calculation"]:
                if verbose:
                    print("Switching to zero point secondary_eclipse which is required for reflected light")
            else:
                shift = shift + 180
        elif zero_point != "secondary_eclipse":
            raise Exception("Do not recognize input zero point. Please specify: night_transit or secondary_eclipse")
        self.inputs["shift"] = shift
        coords, variables = _dataset_like(ds)
        if "temperature" not in variables:
            raise Exception("Must include temperature as data component")
        order = np.argsort(coords["pressure"])
        profs = []
        for i, ph in enumerate(phases):
            lat_f = all_geom[ph]["latitude"] * 180 / np.pi
            lon_f = all_geom[ph]["longitude"] * 180 / np.pi
            # the map as this phase sees it: longitudes advanced by phase + shift, wrapped to [-180, 180)
            lon_rot = (coords["lon"] + ph * 180 / np.pi + shift[i] + 180.0) % 360.0 - 180.0
            rot = dict(coords, lon=lon_rot)
            v = _regrid_lonlat(rot, variables, lon_f, lat_f)
            pr = {"pressure": coords["pressure"][order]}
            for k, a in v.items():
                pr[k] = np.ascontiguousarray(np.transpose(a[:, :, order], (2, 0, 1)))
            profs.append(pr)
        self.inputs["atmosphere"]["profile_4d"] = profs
        self.inputs["atmosphere"]["exclude_mol"] = exclude_mol
        self.nlevel = len(order)

    @_lib.serialized
    def phase_curve(self, opacityclass, full_output=False, plot_opacity=False, n_cpu=1, verbose=False,
                    clouds_by_phase=None, devices=None, options=None):
        """Spectrum at every phase of ``phase_curve_geometry`` (reference justdoit.py:4741-4777; its
        ``n_cpu`` joblib fan-out is a loop here: the phases share the resident opacity tables and one
        GPU).  ``devices=N`` (or a list of device indices) deals the phases out to N GPUs round-robin -- the
        reference's fan-out of whole phases, with a replica of the opacity tables resident on every device
        (``optics.shard_opacity`` over the full grid, uploaded once) and every phase enqueued before the first
        result is read.  Returns ``{phase: spectrum output}``."""
        phases = self.inputs["phase_angle"]
        all_geom = self.inputs["disco"]
        if not isinstance(all_geom, dict) or "calculation" not in all_geom:
            raise Exception("run phase_curve_geometry() first")
        profs = self.inputs["atmosphere"].get("profile_4d")
        if profs is None or len(profs) != len(phases):
            raise Exception("atmosphere_4d() needs one profile per phase (%d)" % len(phases))
        calculation = all_geom["calculation"]
        replicas = [opacityclass]
        if devices is not None:
            devs = _device_list(devices)
            cache = opacityclass.__dict__.setdefault("_replicas", {})
            if tuple(devs) not in cache:
                cache[tuple(devs)] = [optics.shard_opacity(opacityclass, 0, opacityclass.nwno, c)
                                      for c in _device_contexts(devs, opacityclass.ctx)]
            replicas = cache[tuple(devs)]
            for rep in replicas:               # star() / query_method / raman_db may have changed since the copy
                optics.resync_shard(rep, opacityclass, 0, opacityclass.nwno)
        # Every phase is enqueued before the first result is copied back: the GPU runs the phases back to
        # back while the host sets up the next one (one facet-form ATMSETUP and one batched gas stage per
        # phase), and the copies back (each a stream synchronisation) come at the end.  The input planes
        # of a phase return to the context's block cache as soon as its kernels are enqueued (reuse is
        # ordered on the stream); at most `in_flight` phases keep their small result buffers pending.
        opt = _options.current(options)
        in_flight = opt.phases_in_flight
        # One solver launch per leg for a CHUNK of phases (picaso_get_reflected_3d_batch_dev / _thermal_3d_batch_dev;
        # "all phases as one batched launch instead of joblib processes"): the phases' planes
        # stay resident until the chunk's launch, so the chunk is sized to ~48 GB of planes (PICASO_AMD_PHASE_CHUNK
        # overrides; 1 = one launch per phase, the round-3 form).  One batch per replica with devices=N.
        nfac_pc = all_geom[phases[0]]["num_gangle"] * all_geom[phases[0]]["num_tangle"]
        per_phase = 11 * 8.0 * max(self.nlevel - 1, 1) * opacityclass.nwno * nfac_pc
        chunk = opt.phase_chunk or max(1, min(in_flight, int(48e9 // max(per_phase, 1.0))))
        batches = [_SolveBatch() if chunk > 1 else None for _ in replicas]
        results, pending = {}, []
        try:
            for i, ph in enumerate(phases):
                if verbose:
                    print("Currently computing Phase", (i, ph))
                self.inputs["phase_angle"] = ph
                self.inputs["disco"] = all_geom[ph]
                self.inputs["atmosphere"]["profile_3d"] = profs[i]
                if clouds_by_phase is not None:
                    self.inputs["clouds"]["profile_3d"] = clouds_by_phase[i]
                bt = batches[i % len(replicas)]
                pending.append((ph, picaso(self, replicas[i % len(replicas)], dimension="3d",
                                           calculation=calculation, full_output=full_output,
                                           plot_opacity=plot_opacity, defer=True, options=opt, _batch=bt)))
                if bt is not None and bt.pending() >= chunk * (2 if "+" in calculation else 1):
                    bt.flush()
                if len(pending) >= in_flight:
                    for b_ in batches:
                        if b_ is not None:
                            b_.flush()
                    p0, fin = pending.pop(0)
                    results[p0] = fin()
            for b_ in batches:
                if b_ is not None:
                    b_.flush()
            for p0, fin in pending:
                results[p0] = fin()
        finally:
            self.inputs["phase_angle"], self.inputs["disco"] = phases, all_geom
        return results

    def clouds_3d(self, ds=None, regrid=True, plot=True, iz_plot=0, iw_plot=0, verbose=True, df=None):
        """The reference's call (justdoit.py:4515-4620): ``ds`` = cloud ``opd`` / ``w0`` / ``g0`` on
        ``(lon, lat, pressure, wno)`` (layer pressures; xarray-like or dictionary, see ``_dataset_like``), regridded
        onto the facets of ``phase_angle()`` (``regrid=True``) or checked against them, then interpolated in
        wavenumber by ``picaso()`` like any cloud table.  Arrays that are already on the facets are taken as before:
        a dictionary without ``lon`` / ``lat`` (or ``df=``) holding ``opd`` / ``w0`` / ``g0`` as
        ``(nlayer, nwno, num_gangle, num_tangle)``, or ``(nlayer, nwno)`` for a cloud that is the same on every facet."""
        self.inputs["clouds"]["dims"] = "3d"
        if ds is None:
            ds = df
        if ds is None or (isinstance(ds, dict) and not hasattr(ds, "coords") and "lon" not in ds and "lat" not in ds) \
                or not (isinstance(ds, dict) or hasattr(ds, "coords")):
            self.inputs["clouds"]["profile_3d"] = ds
            return
        coords, variables = _dataset_like(ds, extra_coords=("wno",))
        for need, what in (("opd", "optical detph"), ("g0", "assymetry"), ("w0", "single scattering")):
            if need not in variables:
                raise Exception("Must include '%s' (%s) as data component" % (need, what))
        geom = self.inputs["disco"]
        lat_f, lon_f = geom["latitude"] * 180 / np.pi, geom["longitude"] * 180 / np.pi
        if regrid:
            variables = _regrid_lonlat(coords, variables, lon_f, lat_f)
        else:
            assert np.array_equal(lat_f, coords["lat"]) and np.array_equal(lon_f, coords["lon"]), \
                "Cloud latitudes / longitudes do not match the PICASO grid: provide the native grid with regrid=True"
        order = np.argsort(coords["pressure"])
        wno_c = coords["wno"]
        worder = np.argsort(wno_c)
        out = {}
        for k in ("opd", "w0", "g0"):                                   # (lon, lat, p, wno) -> (nlayer, nwno, ng, nt)
            v = variables[k][:, :, order][:, :, :, worder]
            out[k] = np.ascontiguousarray(np.transpose(v, (2, 3, 0, 1)))
        out["wavenumber"] = wno_c[worder]
        out["pressure"] = coords["pressure"][order]
        self.inputs["clouds"]["profile_3d"] = out

    def surface_reflect(self, albedo, wavenumber=None, old_wavenumber=None):
        """Surface reflectivity, scalar or per wavelength (reference justdoit.py:4092)."""
        self.inputs["surface_reflect"] = albedo
        self.inputs["hard_surface"] = 1

    def approx(self, single_phase="TTHG_ray", multi_phase="N=2", delta_eddington=True,
               raman="pollack", tthg_frac=[1, -1, 2], tthg_back=-0.5, tthg_forward=1, p_reference=1,
               rt_method="toon", stream=2, toon_coefficients="quadrature", single_form="explicit",
               calculate_fluxes="off", w_single_form="TTHG", w_multi_form="TTHG", psingle_form="TTHG",
               w_single_rayleigh="on", w_multi_rayleigh="on", psingle_rayleigh="on",
               get_lvl_flux=False):
        """String options -> the integers the solvers take (reference justdoit.py:4635-4738)."""
        if rt_method not in ("toon", "SH"):
            raise Exception("rt_method must be 'toon' or 'SH'")
        if calculate_fluxes not in ("off", "on"):
            raise Exception("calculate_fluxes must be 'off' or 'on'")
        # 'isotropic' (index 2) is accepted as in the reference (justdoit.py:4730-4732); its solver has no branch for it and
        # falls through with the l >= 1 Legendre weights at their initial ones and p_single = 0 (fluxes.py:2805-2855),
        # which the kernels reproduce (tests/golden/sh_extra_*.npz)
        a = self.inputs["approx"]
        a["get_lvl_flux"] = get_lvl_flux
        a["rt_method"] = rt_method
        a["p_reference"] = p_reference
        c = a["rt_params"]["common"]
        c["stream"] = 2 if rt_method == "toon" else int(stream)      # justdoit.py:4702-4705
        if c["stream"] not in (2, 4):
            raise Exception("stream must be 2 or 4")
        c["delta_eddington"] = delta_eddington
        sh = a["rt_params"]["SH"]
        sh["single_form"] = SH_psingle_form_options(False).index(single_form)
        sh["w_single_form"] = SH_scattering_options(False).index(w_single_form)
        sh["w_multi_form"] = SH_scattering_options(False).index(w_multi_form)
        sh["psingle_form"] = SH_scattering_options(False).index(psingle_form)
        sh["w_single_rayleigh"] = SH_rayleigh_options(False).index(w_single_rayleigh)
        sh["w_multi_rayleigh"] = SH_rayleigh_options(False).index(w_multi_rayleigh)
        sh["psingle_rayleigh"] = SH_rayleigh_options(False).index(psingle_rayleigh)
        sh["calculate_fluxes"] = ["off", "on"].index(calculate_fluxes)       # justdoit.py:4736
        c["raman"] = raman_options().index(raman)
        if not isinstance(tthg_frac, (list, np.ndarray)):
            raise Exception("tthg_frac should be a list or ndarray of length=3")
        if len(tthg_frac) != 3:
            raise Exception("tthg_frac should be of length=3 so that : tthg_frac[0] + "
                            "tthg_frac[1]*g_b^tthg_frac[2]")
        c["TTHG_params"].update(fraction=list(tthg_frac), constant_back=tthg_back,
                                constant_forward=tthg_forward)
        t = a["rt_params"]["toon"]
        t["toon_coefficients"] = toon_phase_coefficients(False).index(toon_coefficients)
        t["multi_phase"] = multi_phase_options(False).index(multi_phase)
        t["single_phase"] = single_phase_options(False).index(single_phase)

    def spectrum(self, opacityclass, calculation="reflected", dimension="1d", full_output=False,
                 plot_opacity=False, as_dict=True, devices=None, gather="host", options=None):
        """Run the spectrum (reference justdoit.py:4779-4840).  ``devices=N`` (or a list of device indices):
        the wavelength grid is cut into N contiguous blocks, one per GPU (``picaso(devices=...)``)."""
        if dimension not in ("1d", "3d"):
            raise Exception("dimension must be '1d' or '3d'")
        have = self.inputs["atmosphere"].get("profile" if dimension == "1d" else "profile_3d")
        if have is None:
            raise Exception("Need to set atmosphere profile with the atmosphere%s() function"
                            % ("" if dimension == "1d" else "_3d"))
        if self.inputs["planet"]["gravity"] is None:
            raise Exception("Need to set gravity with the gravity() function")
        return picaso(self, opacityclass, dimension=dimension, calculation=calculation,
                      full_output=full_output, plot_opacity=plot_opacity, as_dict=as_dict, devices=devices,
                      gather=gather, options=options)

    def spectrum_async(self, opacityclass, calculation="reflected", dimension="1d", full_output=False, as_dict=True,
                       options=None):
        """``spectrum()`` without the wait: set-up and every launch of this spectrum are enqueued (result copies
        included) and a ``PendingSpectrum`` comes back at once; its ``result()`` is ``spectrum()``'s dictionary, bit for
        bit.  A retrieval that asks for sample i + 1 before it reads sample i hides the host set-up of one call behind the
        GPU time of the other (``picaso_async``).  No counterpart in the reference, whose callers fan out whole processes
        (justdoit.py:4741-4777)."""
        if dimension not in ("1d", "3d"):
            raise Exception("dimension must be '1d' or '3d'")
        have = self.inputs["atmosphere"].get("profile" if dimension == "1d" else "profile_3d")
        if have is None:
            raise Exception("Need to set atmosphere profile with the atmosphere%s() function"
                            % ("" if dimension == "1d" else "_3d"))
        if self.inputs["planet"]["gravity"] is None:
            raise Exception("Need to set gravity with the gravity() function")
        return picaso_async(self, opacityclass, dimension=dimension, calculation=calculation, full_output=full_output,
                            as_dict=as_dict, options=options)


@_lib.serialized
def picaso(bundle, opacityclass, dimension="1d", calculation="reflected", full_output=False,
           plot_opacity=False, as_dict=True, defer=False, devices=None, gather="host", options=None, _raw=False,
           _shared=None, _batch=None):
    """Spectrum driver (reference ``picaso()``, justdoit.py:65-621).

    The work is ``spectrum.Spectrum``: plan (ATMSETUP, opacity planes) -> enqueue (every leg's solver launches) ->
    finish (results back, integrals, the output dictionary), the same three stages for Toon / SH, 1-D / 3-D,
    monochromatic / correlated-k tables, patchy clouds, level fluxes.  In front of it sit two shortcuts with the same
    results bit for bit: the plain 1-D Toon spectrum goes through ONE C call (``_picaso_driver``, csrc/driver.hip), and
    ``devices=N`` cuts the grid into N wavelength blocks, one per GPU (``_picaso_devices``).

    ``defer=True`` (used by ``phase_curve`` / ``spectrum_batch`` / the block loop of ``devices=``): every kernel of the
    spectrum is enqueued and the ``Spectrum`` is returned; calling it copies the results back and finishes the output
    dictionary -- the caller can enqueue the next spectrum while the GPU is still working on this one.
    ``options``: an ``options.Options`` (A/B and test switches; default: from the environment)."""
    opt = _options.current(options)
    if not (full_output or defer or _raw) and not any(leg in calculation for leg in ("reflected", "thermal", "transmission")):
        # the reference tests `'reflected' in calculation` etc. and nothing else: a string that names no leg runs its
        # set-up and returns the wavenumber grid alone (justdoit.py:254, 318, 388, 517-621) -- no error there, none here
        return {"wavenumber": opacityclass.wno}
    if devices is not None:
        return _picaso_devices(bundle, opacityclass, devices, gather, dimension=dimension, calculation=calculation,
                               full_output=full_output, plot_opacity=plot_opacity, as_dict=as_dict, defer=defer, opt=opt)
    if dimension in ("1d", "3d") and not (full_output or defer or _raw or plot_opacity) and _shared is None and _batch is None:
        fast = _picaso_driver(bundle, opacityclass, [(0, opacityclass.nwno, opacityclass)], calculation, opt, dimension)
        if fast is not None:
            return fast
    s = Spectrum(bundle, opacityclass, dimension=dimension, calculation=calculation, full_output=full_output,
                 as_dict=as_dict, raw=_raw, shared=_shared, batch=_batch, options=opt)
    s.plan().enqueue()
    return s if defer else s.finish()



# ------------------------------------------------------------------------------------------------
# one C call per spectrum (csrc/driver.hip): the launch sequence of the 1-D Toon path for every wavelength block
# ------------------------------------------------------------------------------------------------
def _picaso_driver(bundle, opa, subs, calculation, opt=None, dimension="1d"):
    """The plain 1-D (Toon or SH) or 3-D spectrum through ONE C call (``onecall.run``); None outside what the driver
    covers."""
    from . import onecall
    return onecall.run(bundle, opa, subs, calculation, _options.current(opt), dimension)


# ------------------------------------------------------------------------------------------------
# a spectrum whose result is read later: the host side of call i + 1 runs while the GPU works on call i
# ------------------------------------------------------------------------------------------------
ASYNC_DEPTH = 4        # spectra of one opacity object in flight through the C driver (a block table -- planes, pinned result
#                        blocks -- per slot; a fifth call first finishes the oldest)


class PendingSpectrum:
    """Handle of ``picaso_async`` / ``inputs.spectrum_async``: ``result()`` (or calling it) waits for this spectrum's result
    copies, forms the output dictionary -- ``spectrum()``'s, bit for bit -- and returns it (the same object every time)."""

    def __init__(self, finish, abandon=None):
        self._finish, self._abandon, self._out, self._err = finish, abandon, None, None

    def done(self):
        return self._finish is None

    @_lib.serialized
    def result(self):
        if self._finish is not None:
            fin, self._finish = self._finish, None
            try:
                self._out = fin()
            except BaseException as exc:         # the launches are on the stream: drop their result copies, keep the error
                self._err = exc
                if self._abandon is not None:
                    self._abandon()
            self._abandon = None
        if self._err is not None:
            raise self._err
        return self._out

    __call__ = result


@_lib.serialized
def picaso_async(bundle, opacityclass, dimension="1d", calculation="reflected", full_output=False, as_dict=True,
                 options=None):
    """``picaso()`` up to and including the last launch; returns a ``PendingSpectrum``.

    What the C driver covers (``onecall.prepare``: the plain 1-D Toon / SH and 3-D spectra) is enqueued through it on one of
    ``ASYNC_DEPTH`` block tables of the opacity object -- each with its own planes and pinned result blocks, so the spectra
    in flight do not touch each other's memory; the streams order them.  Everything else (k-tables mixed on the fly,
    patchy clouds, level fluxes, ``full_output`` ...) is ``picaso(defer=True)`` with its result copies already on the
    stream.  Either way the dictionary is the one ``picaso()`` returns for the same inputs.  The case must not be given a
    new star before ``result()`` (the flux ratios are formed there); atmosphere, clouds and geometry may change at once."""
    from . import onecall
    from . import driver as drv
    opt = _options.current(options)
    legs = any(leg in calculation for leg in ("reflected", "thermal", "transmission"))
    if not full_output and not legs:
        out = {"wavenumber": opacityclass.wno}
        return PendingSpectrum(lambda: out)
    if not full_output:
        st = opacityclass.__dict__.setdefault("_async_slots", {"n": 0, "live": {}})
        slot = st["n"] % ASYNC_DEPTH
        old = st["live"].pop(slot, None)
        if old is not None and not old.done():
            try:
                old.result()                      # the slot's planes and pinned blocks are about to be reused
            except BaseException:
                pass                              # its owner gets the error from result()
        subs = [(0, opacityclass.nwno, opacityclass)]
        p = onecall.prepare_3d(bundle, opacityclass, subs, calculation, opt, slot=slot) if dimension == "3d" else \
            onecall.prepare(bundle, opacityclass, subs, calculation, opt, slot=slot, early=True)
        if p is not None:
            p["inp"] = dict(p["inp"], star=dict(p["inp"]["star"]))      # what finish() reads of the case, as it is now
            try:
                drv.enqueue(p["table"], p["job"], p.get("phase", 0))
            except BaseException:
                drv.abandon(p["table"])
                raise
            h = PendingSpectrum(lambda: onecall.finish(p), lambda: drv.abandon(p["table"]))
            st["live"][slot] = h
            st["n"] += 1
            return h
    s = picaso(bundle, opacityclass, dimension=dimension, calculation=calculation, full_output=full_output, as_dict=as_dict,
               defer=True, options=opt)
    if not opt.sync_copies:
        s.prefetch()
    return PendingSpectrum(s)


# ------------------------------------------------------------------------------------------------
# several spectra in one launch: the retrieval / grid callers of the reference run
# spectrum() once per sample in separate processes (driver.py:405-426, justdoit.py:4741-4777)
# ------------------------------------------------------------------------------------------------
class _SolveBatch:
    """The Toon solver launches of several ``picaso(defer=True, _batch=...)`` calls, issued by ``flush()`` as ONE
    batched launch per group of spectra that share shape and options (``picaso_get_reflected_1d_batch_dev`` /
    ``picaso_get_thermal_1d_batch_dev``): every spectrum is bit-identical to its own launch, the GPU sees one grid
    that fills it instead of B that each leave its SIMDs half empty (DESIGN.md sections 5 and 6)."""

    def __init__(self):
        self.refl, self.therm, self.refl3, self.therm3 = {}, {}, {}, {}

    def add_reflected(self, key, item):
        self.refl.setdefault(key, []).append(item)

    def add_thermal(self, key, item):
        self.therm.setdefault(key, []).append(item)

    def add_reflected_3d(self, key, item):
        self.refl3.setdefault(key, []).append(item)

    def add_thermal_3d(self, key, item):
        self.therm3.setdefault(key, []).append(item)

    def pending(self):
        return sum(len(v) for d in (self.refl, self.therm, self.refl3, self.therm3) for v in d.values())

    def flush(self):
        for key, items in self.refl3.items():
            nlevel, nwno, ng, nt, tt, present, gw, tw = key
            fn = resident.reflected_3d_fm_batch if "facet-major" in present else resident.reflected_3d_batch
            fn(items[0]["ctx"], nlevel, nwno, ng, nt, [it["planes"] for it in items],
                                        [it["rs"] for it in items],
                                        np.stack([np.asarray(it["ubar0"], dtype=float).reshape(ng, nt) for it in items]),
                                        np.stack([np.asarray(it["ubar1"], dtype=float).reshape(ng, nt) for it in items]),
                                        np.array([it["cos_theta"] for it in items], dtype=float),
                                        [it["F0PI"] for it in items], *tt, [it["xint"] for it in items], gweight=gw,
                                        tweight=tw, albedo=[it["albedo"] for it in items])
        for key, items in self.therm3.items():
            nlevel, nwno, ng, nt, hard, _, (has_g, fm), gw, tw = key
            fn = resident.thermal_3d_fm_batch if fm else resident.thermal_3d_batch
            fn(items[0]["ctx"], nlevel, items[0]["wno"], nwno, ng, nt,
                                      np.stack([it["tlevel"] for it in items]), [it["dtau"] for it in items],
                                      [it["w0"] for it in items], [it["cosb"] for it in items] if has_g else None,
                                      np.stack([it["plevel"] for it in items]),
                                      np.stack([np.asarray(it["ubar1"], dtype=float).reshape(ng, nt) for it in items]),
                                      [it["rs"] for it in items], hard, [it["flux"] for it in items], gweight=gw,
                                      tweight=tw, flux_disk=[it["disk"] for it in items])
        self.refl3, self.therm3 = {}, {}
        for key, items in self.refl.items():
            nlevel, nwno, ng, nt, tt, tcoef, b_top, gw, tw, _ = key
            ctx = items[0]["ctx"]
            if len(items) == 1 or ng * nt > 8:       # the batched launch carries at most 8 disk angles: a 6 x 6 disk goes alone
                for it in items:
                    _reflected(ctx, nlevel, nwno, ng, nt, it["planes"], it["rs"], it["ubar0"], it["ubar1"],
                               it["cos_theta"], it["F0PI"], *tt, tcoef, b_top, it["xint"], None, gw, tw, it["albedo"])
                continue
            same = all(np.array_equal(it["ubar0"], items[0]["ubar0"]) and np.array_equal(it["ubar1"], items[0]["ubar1"])
                       and it["cos_theta"] == items[0]["cos_theta"] for it in items)
            u0 = items[0]["ubar0"] if same else np.stack([np.asarray(it["ubar0"], dtype=float).reshape(ng, nt) for it in items])
            u1 = items[0]["ubar1"] if same else np.stack([np.asarray(it["ubar1"], dtype=float).reshape(ng, nt) for it in items])
            ct = items[0]["cos_theta"] if same else np.array([it["cos_theta"] for it in items], dtype=float)
            resident.reflected_1d_batch(ctx, nlevel, nwno, ng, nt, [it["planes"] for it in items],
                                        [it["rs"] for it in items], u0, u1, ct, [it["F0PI"] for it in items], *tt,
                                        [it["xint"] for it in items], toon_coefficients=tcoef, b_top=b_top,
                                        gweight=gw, tweight=tw, albedo=[it["albedo"] for it in items])
        for key, items in self.therm.items():
            nlevel, nwno, ng, nt, hard, _, gw, tw = key
            ctx = items[0]["ctx"]
            if len(items) == 1 or ng * nt > 8:
                for it in items:
                    resident.thermal_1d(ctx, nlevel, it["wno"], nwno, ng, nt, it["tlevel"], it["dtau"], it["w0"], it["cosb"],
                                        it["plevel"], it["ubar1"], it["rs"], hard, it["flux"], gweight=gw, tweight=tw,
                                        flux_disk=it["disk"])
                continue
            same = all(np.array_equal(it["ubar1"], items[0]["ubar1"]) for it in items)
            u1 = items[0]["ubar1"] if same else np.stack([np.asarray(it["ubar1"], dtype=float).reshape(ng, nt) for it in items])
            resident.thermal_1d_batch(ctx, nlevel, items[0]["wno"], nwno, ng, nt, np.stack([it["tlevel"] for it in items]),
                                      [it["dtau"] for it in items], [it["w0"] for it in items],
                                      [it["cosb"] for it in items], np.stack([it["plevel"] for it in items]), u1,
                                      [it["rs"] for it in items], hard, [it["flux"] for it in items], gweight=gw,
                                      tweight=tw, flux_disk=[it["disk"] for it in items])
        self.refl, self.therm = {}, {}


@_lib.serialized
def spectrum_batch(cases, opacityclass, calculation="reflected", full_output=False, as_dict=True, batch_size=4,
                   options=None):
    """``[case.spectrum(opacityclass, calculation) for case in cases]`` (1-D) with the solvers of up to
    ``batch_size`` spectra in ONE launch each: what a retrieval or a model grid asks of the reference one
    ``spectrum()`` call and one process at a time (driver.py:405-426).  ``cases``: ``inputs`` objects, each with its
    own atmosphere / clouds / geometry / approximations; spectra whose grids and options agree share a launch (the
    others go alone), correlated-k, SH, patchy-cloud and level-flux cases take their usual path.  Every output
    dictionary is bit-identical to ``case.spectrum(...)``'s.  The chunks are pipelined: while the GPU solves chunk k the
    host sets up and enqueues chunk k + 1, and only then waits for k's result copies (``picaso_memcpy_d2h_async`` into
    pinned blocks, ``picaso_mark_wait``) and runs its integrals.  HBM: the planes of a chunk stay resident until its
    launch (0.8 GB per cloudy 1e5 x 90 spectrum, 0.2 GB per cloud-free one), two chunks at a time."""
    opt = _options.current(options)
    cases = list(cases)
    outs = []
    in_flight = []                 # chunks whose launches and result copies are on the stream
    B = max(1, int(batch_size))
    for c0 in range(0, len(cases), B):
        chunk = cases[c0:c0 + B]
        for case in chunk:
            if case.inputs["atmosphere"].get("profile") is None:
                raise Exception("Need to set atmosphere profile with the atmosphere() function")
            if case.inputs["planet"]["gravity"] is None:
                raise Exception("Need to set gravity with the gravity() function")
        batch = _SolveBatch()
        fins = []
        for case in chunk:
            fins.append(picaso(case, opacityclass, dimension="1d", calculation=calculation, full_output=full_output,
                               as_dict=as_dict, defer=True, options=opt, _batch=batch))
        batch.flush()
        # integrals and result copies on a stream of their own, behind this chunk's solvers
        post = {}
        for fin in fins:
            key = getattr(fin.ctx, "value", fin.ctx)
            if key not in post and not opt.no_post_stream:
                pc = _lib.aux_context(_lib.device_of(fin.ctx), 1 << 20)
                _lib.ctx_wait(pc, fin.ctx)
                if fin.tctx is not fin.ctx:
                    _lib.ctx_wait(pc, fin.tctx)
                post[key] = pc
            fin.prefetch(post.get(key))
        # the host finishes chunk k (waits for its copies, integrals, ratios) only after chunk k + 1 has been set up and
        # enqueued: the GPU solves k + 1 meanwhile, and set-up of k + 1 ran while it solved k
        in_flight.append(fins)
        if len(in_flight) > 1:
            outs.extend(fin() for fin in in_flight.pop(0))
    for fins in in_flight:
        outs.extend(fin() for fin in fins)
    return outs


# ------------------------------------------------------------------------------------------------
# one spectrum on several GPUs; replaces the reference's process fan-out, justdoit.py:4741-4777
# ------------------------------------------------------------------------------------------------
_extra_ctx = {}


def _device_list(devices):
    devs = list(range(int(devices))) if isinstance(devices, (int, np.integer)) else [int(d) for d in devices]
    ndev = _lib.device_count()
    if not devs or min(devs) < 0:
        raise Exception("devices must be a positive count or a list of device indices, got %r" % (devices,))
    need = max(devs) + 1
    if need > ndev:
        raise _lib.PicasoHipError("devices=%r needs %d GPU(s), %d visible" % (devices, need, ndev))
    return devs


def _device_contexts(devs, home_ctx):
    """One context per entry of ``devs``.  The first entry of a device is the process's context on it (the
    opacity object's own context where that is its device); a device listed again gets a further context (its
    own stream) -- how the sharded path is exercised on a single GPU."""
    import os as _os
    home_dev = _lib.device_of(home_ctx)
    seen, out = {}, []
    for d in devs:
        k = seen.get(d, 0)
        seen[d] = k + 1
        if k == 0:
            out.append(home_ctx if d == home_dev else _lib.context(d))
        else:
            key = (_os.getpid(), d, k)
            if key not in _extra_ctx:
                _extra_ctx[key] = _lib.new_context(d)
            out.append(_extra_ctx[key])
    return out


def _opacity_shards(opa, devs):
    """[(lo, hi, shard)]: the wavelength blocks of ``sharding.shard_bounds`` with their tables resident on the
    devices of ``devs``; built once per (opacity object, device list) and kept on the object."""
    from . import sharding
    cache = opa.__dict__.setdefault("_shards", {})
    key = tuple(devs)
    if key not in cache:
        ctxs = _device_contexts(devs, opa.ctx)
        bounds = sharding.shard_bounds(opa.nwno, len(devs))
        if bounds[-1][1] - bounds[-1][0] < 1:
            raise Exception("devices=%d but the grid has only %d wavelengths" % (len(devs), opa.nwno))
        cache[key] = [(lo, hi, optics.shard_opacity(opa, lo, hi, c)) for (lo, hi), c in zip(bounds, ctxs)]
    for lo, hi, sh in cache[key]:          # star() / query_method / raman_db may have changed since the cut
        optics.resync_shard(sh, opa, lo, hi)
    return cache[key]


class _Bundle:
    def __init__(self, inputs_dict, nlevel):
        self.inputs, self.nlevel = inputs_dict, nlevel


def _slice_inputs(inp, lo, hi, nwno, nlayer, clouds=True):
    """The run configuration as one wavelength block sees it: per-wavelength inputs (stellar flux, surface
    reflectivity, cloud tables already on the opacity grid) cut to ``[lo, hi)``; everything else shared.
    ``clouds=False``: the block takes its cloud columns from an atmosphere that was set up once for the whole grid
    (``_atmosphere_block``: views, no copies of the (nlayer, nwno) tables)."""
    out = dict(inp)
    star = dict(inp["star"])
    rf = star.get("relative_flux")
    if isinstance(rf, np.ndarray) and rf.shape == (nwno,):
        star["relative_flux"] = np.ascontiguousarray(rf[lo:hi])
    out["star"] = star
    sr = inp.get("surface_reflect")
    if sr is not None and np.size(sr) == nwno and nwno > 1:
        out["surface_reflect"] = np.ascontiguousarray(np.asarray(sr, dtype=float).reshape(nwno)[lo:hi])
    cl = dict(inp["clouds"])
    prof = cl.get("profile") if clouds else None
    if prof is not None:
        new = {}
        for k in ("opd", "g0", "w0"):
            v = np.asarray(prof[k], dtype=np.float64)
            if v.ndim != 0 and v.size != nlayer and v.size // nlayer == nwno:
                v = np.ascontiguousarray(v.reshape(nlayer, nwno)[:, lo:hi])
            new[k] = v                 # other tables are regridded onto the block's own wavenumbers (get_clouds)
        cl["profile"] = new
    p3 = cl.get("profile_3d")
    if p3 is not None and p3.get("wavenumber") is None:
        # arrays already on the opacity grid: this block's columns.  Tables on a wavenumber grid of their own go to every
        # block as they are (the same dictionary: its resident copies are kept with it) and are interpolated onto the
        # block's wavenumbers there
        cl["profile_3d"] = {k: np.ascontiguousarray(np.asarray(v, dtype=float).reshape((nlayer, nwno, -1))[:, lo:hi])
                            for k, v in p3.items()}
    out["clouds"] = cl
    return out


# full_output entries that carry a wavelength axis (atmsetup.as_dict): name -> axis (None: the last one)
_WAVE_KEYS = {"wavenumber": 0, "w0": 1, "g0": 1, "opd": 1, "taugas": 1, "tauray": 1, "taucld": 1,
              "albedo_3d": None, "thermal_3d": None, "flux_layers": None, "flux_minus": None, "flux_plus": None,
              "flux_minus_mdpt": None, "flux_plus_mdpt": None}


def _merge_blocks(parts, key=None):
    """Join the per-block pieces of a ``full_output`` dictionary: dictionaries recursively, the arrays that carry
    a wavelength axis along it; everything else (profiles, units, geometry) is the same in every block and is
    taken from the first."""
    first = parts[0]
    if isinstance(first, dict):
        return {k: _merge_blocks([p[k] for p in parts], k) for k in first}
    if isinstance(first, np.ndarray) and first.ndim > 0 and key in _WAVE_KEYS:
        ax = _WAVE_KEYS[key]
        return np.concatenate(parts, axis=first.ndim - 1 if ax is None else ax)
    return first


def _picaso_devices(bundle, opa, devices, gather, dimension, calculation, full_output, plot_opacity, as_dict, defer,
                    opt=None):
    """``picaso(..., devices=N)``: ONE spectrum in N contiguous wavelength blocks (``sharding.shard_bounds``), one
    per GPU.  Each block's opacity tables live on its device (``optics.shard_opacity``, uploaded once per
    opacity object), its gas stage, ``compute_opacity`` and solvers run there, all blocks are enqueued before the
    first result is read, and the spectrum-wide integrals (Bond albedo, effective temperature) run on the
    gathered arrays through the same code as the single-GPU path -- every function on the path is pointwise in
    wavelength, so the result is bit-identical to ``devices=None`` (tests/test_devices_gpu.py).
    ``gather='host'``: every device copies its block back (N small copies, the result is needed on the host
    once); ``gather='rccl'``: the blocks are all-gathered over xGMI inside the library (``picaso_comm_init_all`` +
    ``picaso_all_gather_group_dev``: the full spectrum ends up resident on every device) and read from the first.
    Replaces the reference's joblib fan-out (justdoit.py:4741-4777)."""
    from . import sharding
    if gather not in ("host", "rccl"):
        raise Exception("gather must be 'host' or 'rccl'")
    if full_output and not as_dict:
        raise Exception("devices=N returns the merged full_output dictionary: use as_dict=True")
    inp = bundle.inputs
    devs = _device_list(devices)
    shards = _opacity_shards(opa, devs)
    nwno = opa.nwno
    if dimension in ("1d", "3d") and gather == "host" and not (full_output or defer or plot_opacity):
        fast = _picaso_driver(bundle, opa, shards, calculation, opt, dimension)  # every block in one C call (csrc/driver.hip)
        if fast is not None:
            return fast
    nlevel = getattr(bundle, "nlevel", None)
    nlayer = (nlevel - 1) if nlevel else 0
    # The atmosphere set-up (hydrostatic altitude, column densities, cloud tables on the opacity grid) and the table
    # rows / weights of every layer do not depend on the wavelength block: done once for the whole grid, not once
    # per device (0.5 ms of host time per block otherwise, which on eight GPUs is more than the spectrum itself)
    shared = None
    if dimension == "1d" and len(shards) > 1:
        atm0 = _setup_atmosphere(inp, opa, opa.wno)
        plan = None
        if not getattr(opa, "on_fly", False):       # on-the-fly mixing leaves a per-block table on each device
            opa.get_opacities(atm0, exclude_mol=inp["atmosphere"]["exclude_mol"])
            plan = opa._plan
        shared = dict(atm=atm0, plan=plan)
    elif (dimension == "3d" and len(shards) > 1 and opa.ngauss == 1 and not _options.current(opt).facet_loop
          and not inp["approx"].get("get_lvl_flux", False)):
        # 3-D: the facet-form ATMSETUP and the tall plan (rows / weights / coefficients of every (facet, layer)) once
        geom = inp["disco"]
        atm_f, atm0, tlev3, plev3 = setup_facets_3d(inp, opa, opa.wno, geom["num_gangle"], geom["num_tangle"])
        tall = getattr(atm_f, "_fast_tall", None)
        shared = dict(atm_f=atm_f, atm=atm0, tlev3=tlev3, plev3=plev3, tall=tall[:2] if tall is not None else None)
    fins = []
    for lo, hi, sub in shards:
        b = _Bundle(_slice_inputs(inp, lo, hi, nwno, nlayer, clouds=shared is None), nlevel)
        sh = dict(shared, lo=lo, hi=hi) if shared is not None else None
        fins.append(picaso(b, sub, dimension=dimension, calculation=calculation, full_output=full_output,
                           plot_opacity=plot_opacity, as_dict=True, defer=True, options=opt, _raw=True, _shared=sh))
    gathered = {}
    if gather == "rccl" and len(devs) > 1:
        if len(set(devs)) != len(devs):
            raise Exception("gather='rccl' takes every device at most once (RCCL: one rank per device)")
        group = sharding.device_group(devs)
        for f, gctx in zip(fins, group.ctxs):
            if f.tctx is not f.ctx:                      # a leg that ran on the block's second stream
                _lib.ctx_wait(f.ctx, f.tctx)
            # the group posts device i's collective on the process's default context of that device; a block that
            # ran on another context (an opacity object built on new_context()) is ordered in front of it
            if getattr(gctx, "value", gctx) != getattr(f.ctx, "value", f.ctx):
                _lib.ctx_wait(gctx, f.ctx)
        for key in fins[0].dev:
            fulls = [DeviceArray((nwno,), f.ctx) for f in fins]
            group.all_gather_spectrum([f.dev[key] for f in fins], fulls, nwno)
            gathered[key] = fulls

    def finish():
        raws = [f() for f in fins]
        raw = {}
        for key in ("albedo", "thermal", "transit_depth"):
            if key in raws[0]:
                raw[key] = np.concatenate([r[key] for r in raws])
                if key in gathered:
                    full = gathered[key][0].to_host()
                    if not np.array_equal(full, raw[key], equal_nan=True):
                        raise _lib.PicasoHipError("devices=%r: the RCCL-gathered %s differs from the blocks the "
                                                  "devices computed" % (devices, key))
                    raw[key] = full
        if inp["star"]["database"] == "nostar":
            F0PI = np.zeros(nwno) + 1.0
        else:
            F0PI = inp["star"]["relative_flux"]
        stellar = getattr(opa, "unshifted_stellar_spec", None)
        if stellar is None:
            stellar = F0PI
        out = _postprocess(raw, opa.wno, stellar, inp["star"]["semi_major"], inp["star"]["radius"],
                           inp["planet"]["radius"], opa)
        if full_output:
            out["full_output"] = _merge_blocks([r["full_output"] for r in raws])
        return out
    return finish if defer else finish()

