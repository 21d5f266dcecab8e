"""Minimal atmosphere state for the accelerated path (counterpart of the reference ``ATMSETUP``,
picaso/atmsetup.py:17-876, restricted to what ``picaso()`` needs to feed the opacity and solver
kernels).  Host-side numpy only; everything is cgs (no astropy units).

Kept from the reference: level -> layer averaging (atmsetup.py:219-229), molecular weights from
main-isotope masses (``get_weights`` :285-342) and mean molecular weight (:345-361), hydrostatic
altitude and the column density with its half-gravity end layers (:384-461, :549-556), the
cloud-free default and the ``(nlayer, nwno)`` reshape of cloud tables (:558-627),
``get_needed_continuum`` (:248-283).  Pinned to the reference by tests/golden/altitude.npz.
Out of scope: chemistry, 3-D regridding, virga clouds.
"""
import numpy as np


class CloudTables(dict):
    """``layer['cloud']`` for cloud tables that come on their own wavenumber grid (the 196-point grid of virga and of
    the box clouds, justdoit.py:4235-4266): ``opd`` / ``w0`` / ``g0`` as ``(nlayer, nwno)`` arrays on the opacity
    grid, formed with the reference's row-by-row ``numpy.interp`` (wavelength.py:46-70 from atmsetup.py:609-622) only
    when somebody reads them on the host (``full_output``).  ``compute_opacity`` does not: it takes ``compact`` /
    ``in_wno`` / ``wno`` and regrids on the device (``picaso_regrid_rows_dev``, same bits) -- the host regrid costs
    0.5 s per table at 1e5 wavelengths, and the three (nlayer, nwno) tables another 216 MB of host-to-device copy."""

    def __init__(self, compact, in_wno, wno):
        super().__init__()
        self.compact = compact                        # name -> (nlayer, nin) float64
        self.in_wno = np.ascontiguousarray(in_wno, dtype=np.float64)
        self.wno = wno

    def __missing__(self, k):
        if k not in self.compact:
            raise KeyError(k)
        x = np.asarray(self.wno, dtype=np.float64)
        v = np.ascontiguousarray(np.stack([np.interp(x, self.in_wno, row) for row in self.compact[k]]))
        self[k] = v
        return v

    def __iter__(self):
        return iter(self.compact)

    def __len__(self):
        return len(self.compact)

    def __contains__(self, k):
        return k in self.compact

    def keys(self):
        return self.compact.keys()

    def values(self):
        return [self[k] for k in self.compact]

    def items(self):
        return [(k, self[k]) for k in self.compact]

    def columns(self, lo, hi):
        """The tables of the wavelength block ``[lo, hi)`` (interpolation is pointwise in the output grid)."""
        return CloudTables(self.compact, self.in_wno, self.wno[lo:hi])


# Mass (u) of the most abundant isotope of every element: the reference weighs a molecule with these,
# not with standard atomic weights (``get_weights``, atmsetup.py:285-342: argmax of the isotope
# abundances), so H2 is 2.01565 rather than 2.01588 -- it enters every opacity through colden/mmw.
_MAIN_ISOTOPE = {
    "H": 1.0078250321, "D": 2.014101778, "He": 4.0026032497, "Li": 7.016004, "Be": 9.0121821,
    "B": 11.0093055, "C": 12.0, "N": 14.0030740052, "O": 15.9949146221, "F": 18.9984032,
    "Ne": 19.9924401759, "Na": 22.98976967, "Mg": 23.9850419, "Al": 26.98153844, "Si": 27.9769265327,
    "P": 30.97376151, "S": 31.97207069, "Cl": 34.96885271, "Ar": 39.962383123, "K": 38.9637069,
    "Ca": 39.9625912, "Sc": 44.9559102, "Ti": 47.9479471, "V": 50.9439637, "Cr": 51.9405119,
    "Mn": 54.9380496, "Fe": 55.9349421, "Co": 58.9332002, "Ni": 57.9353479, "Cu": 62.9296011,
    "Zn": 63.9291466, "Ga": 68.925581, "Ge": 73.9211782, "As": 74.9215964, "Se": 79.9165218,
    "Br": 78.9183376, "Kr": 83.911507, "Rb": 84.9117893, "Sr": 87.9056143, "Y": 88.9058479,
    "Zr": 89.9047037, "Nb": 92.9063775, "Mo": 97.9054078, "Tc": 97.907216, "Ru": 101.9043495,
    "Rh": 102.905504, "Pd": 105.903483, "Ag": 106.905093, "Cd": 113.9033581, "In": 114.903878,
    "Sn": 119.9021966, "Sb": 120.903818, "Te": 129.9062228, "I": 126.904468, "Xe": 131.9041545,
    "Cs": 132.905447, "Ba": 137.905241, "La": 138.906348, "Ce": 139.905434, "Pr": 140.907648,
    "Nd": 141.907719, "Pm": 144.912744, "Sm": 151.919728, "Eu": 152.921226, "Gd": 157.924101,
    "Tb": 158.925343, "Dy": 163.929171, "Ho": 164.930319, "Er": 165.93029, "Tm": 168.934211,
    "Yb": 173.9388581, "Lu": 174.9407679, "Hf": 179.9465488, "Ta": 180.947996, "W": 183.9509326,
    "Re": 186.9557508, "Os": 191.961479, "Ir": 192.962924, "Pt": 194.964774, "Au": 196.966552,
    "Hg": 201.970626, "Tl": 204.974412, "Pb": 207.976636, "Bi": 208.980383, "Po": 208.982416,
    "At": 209.987131, "Rn": 222.0175705, "Fr": 223.0197307, "Ra": 226.0254026, "Ac": 227.027747,
    "Th": 232.0380504, "Pa": 231.0358789, "U": 238.0507826, "Np": 237.0481673, "Pu": 244.064198,
    "Am": 243.0613727, "Cm": 247.070347, "Bk": 247.070299, "Cf": 251.07958, "Es": 252.08297,
    "Fm": 257.095099, "Md": 258.098425, "No": 259.10102, "Lr": 262.10969, "Rf": 261.10875,
    "Db": 262.11415, "Sg": 266.12193, "Bh": 264.12473, "Hs": 269.13411, "Mt": 268.13882}


_WEIGHT_CACHE = {}


def molecular_weight(name):
    """Cached ``_molecular_weight`` (a pure function of the name; a retrieval asks for the same half-dozen molecules
    in every spectrum)."""
    try:
        return _WEIGHT_CACHE[name]
    except KeyError:
        w = _WEIGHT_CACHE[name] = _molecular_weight(name)
        return w


def _molecular_weight(name):
    """Molecular weight from a formula such as 'H2O', 'CH4', 'TiO', 'CH3D' (case-sensitive elements),
    tokenised as the reference does (``separate_molecule_name`` / ``separate_string_number``,
    atmsetup.py:807-820): charges are ignored ('H3+' weighs 3 H).  Raises ``KeyError`` for anything
    that is not a formula; the reference's '_'-separated isotopologue names are not built."""
    import re
    if name == "e-":
        return 0.0                                            # no element token: weight 0 (:309, :340)
    if "_" in name:
        raise KeyError(name)
    total = 0.0
    tokens = re.findall(r"[A-Z][a-z]?\d*|\d+", name)
    if not tokens:
        raise KeyError(name)
    for tok in tokens:
        sep = re.findall(r"[A-Za-z]+|\d+", tok)
        el, num = (sep[0], 1) if len(sep) == 1 else sep
        total += _MAIN_ISOTOPE[el] * float(num)
    return total


class _Consts:
    pconv = 1e6                 # bar -> dyn/cm2 (atmsetup.py:50)
    k_b = 1.380649e-16          # erg/K
    G = 6.6743e-8
    amu = 1.66053906660e-24     # g
    rgas = 8.31446261815324     # J/K/mol, as the reference's c.R.value (atmsetup.py:56)
    pi = np.pi


class _Obj:
    pass


class ATMSETUP:
    def __init__(self, config):
        self.input = config
        self.warnings = []
        self.c = _Consts()
        self.planet = _Obj()
        self.planet.radius = self.planet.mass = np.nan       # gravity-only planet unless the caller sets them
        self.layer, self.level = {}, {}
        self.dimension = "1d"

    def add_warnings(self, w):
        self.warnings += [w]

    def get_profile(self):
        """Level columns -> level / layer state.  Columns are ``(nlevel,)`` arrays, or -- the facet form
        the 3-D path uses -- ``(nlevel, nfacets)`` for the facet-dependent ones and ``(nlevel, 1)`` for
        the shared ones: every method below slices along axis 0 only, so one ATMSETUP then carries all
        facets at once (64 facet set-ups per 3-D spectrum become one)."""
        read = self.input["atmosphere"]["profile"]           # dict-like / DataFrame of level columns
        cols = list(read.keys())
        weights, molecules = {}, []
        for k in cols:
            if k in ("pressure", "temperature"):
                continue
            if k == "e-":
                self.level["electrons"] = np.asarray(read["e-"], dtype=float)
                self.layer["electrons"] = 0.5 * (self.level["electrons"][1:] + self.level["electrons"][:-1])
                continue
            try:
                weights[k] = molecular_weight(k)
                molecules.append(k)
            except KeyError:
                self.add_warnings("Ignoring %s in input file, not recognized molecule" % k)
        self.weights = weights
        self.molecules = np.array(molecules, dtype=str)
        self.level["mixingratios"] = {m: np.asarray(read[m], dtype=float) for m in molecules}
        self.layer["mixingratios"] = {m: 0.5 * (v[1:] + v[:-1])
                                      for m, v in self.level["mixingratios"].items()}
        self.level["temperature"] = np.asarray(read["temperature"], dtype=float)
        self.level["pressure_bar"] = np.asarray(read["pressure"], dtype=float)
        self.level["pressure"] = self.level["pressure_bar"] * self.c.pconv
        self.layer["temperature"] = 0.5 * (self.level["temperature"][1:] + self.level["temperature"][:-1])
        self.layer["pressure"] = np.sqrt(self.level["pressure"][1:] * self.level["pressure"][:-1])
        self.c.nlevel = len(self.level["temperature"])
        self.c.nlayer = self.c.nlevel - 1

    def get_mmw(self):
        w = 0.0                                               # 0 + x = x exactly: as np.zeros(nlevel) + ...
        for m in self.molecules:
            w = w + self.level["mixingratios"][m] * self.weights[m]
        self.level["mmw"] = w
        self.layer["mmw"] = 0.5 * (w[:-1] + w[1:])

    def get_density(self):
        self.level["den"] = self.level["pressure"] / (self.c.k_b * self.level["temperature"])

    def get_dtdp(self):
        """d ln T / d ln P per layer (reference atmsetup.py:371-383)."""
        self.layer["dtdp"] = (np.diff(np.log(self.level["temperature"])) /
                              np.diff(np.log(self.level["pressure"])))

    def get_altitude(self, p_reference=1, constant_gravity=False):
        """Level altitude ``z``, thickness ``dz`` and gravity by hydrostatic integration outwards
        from the reference pressure (reference atmsetup.py:384-461).  Without a planet radius the
        gravity is constant (``:398``) and ``z`` is NaN, as in the reference.  The layer gravity is
        the mean of the level values *before* the two end levels are filled in (``:453``), so the
        top and bottom layers carry half the gravity -- kept, it sets ``colden`` and every opacity."""
        c, planet = self.c, self.planet
        p_reference = p_reference * c.pconv
        if np.isnan(planet.radius):
            constant_gravity = True
        mmw = self.level["mmw"] * c.amu
        tlevel, plevel = self.level["temperature"], self.level["pressure"]
        if p_reference >= np.max(plevel):
            p_reference = np.max(plevel)
        else:
            p_reference = plevel[plevel >= p_reference][0]    # snap to the pressure grid (:414)
        shape = np.broadcast(tlevel, mmw, plevel).shape       # (nlevel,) or, facet form, (nlevel, nfacets)
        z = np.zeros(shape) + planet.radius
        dz = np.zeros(shape)
        gravity = np.zeros(shape)
        n = len(plevel)
        if constant_gravity and n > 1 and np.all(np.diff(np.reshape(plevel, (n, -1))[:, 0]) > 0):
            # same arithmetic as the level loops below, element-wise (gravity does not depend on z; the running sums
            # are sequential like the loops: cumsum along the level axis).  1-D, or the facet form -- (nlevel, nfacets)
            # temperatures over a shared (nlevel, 1) pressure column: 64 facets x 91 levels per 3-D spectrum
            g = planet.gravity
            iref = int(np.argmax(np.reshape(plevel, (n, -1))[:, 0] >= p_reference))
            scale_h = np.broadcast_to(c.k_b * tlevel / (mmw * g), shape)
            if iref < n - 1:                                  # inwards from the reference level
                gravity[iref:n - 1] = g
                dz[iref:n - 1] = scale_h[iref:n - 1] * np.log(plevel[iref + 1:] / plevel[iref:n - 1])
                z[iref:] = np.cumsum(np.concatenate((z[iref][None], -dz[iref:n - 1]), axis=0), axis=0)
            if iref >= 1:                                     # outwards
                gravity[1:iref + 1] = g
                dz[1:iref + 1] = scale_h[1:iref + 1] * np.log(plevel[1:iref + 1] / plevel[0:iref])
                z[iref::-1] = np.cumsum(np.concatenate((z[iref][None], dz[iref:0:-1]), axis=0), axis=0)
            return self._finish_altitude(z, dz, gravity, lambda i: g, tlevel, mmw)

        def g_at(i):
            if constant_gravity:
                return planet.gravity
            zi = z[i]
            if np.ndim(zi):
                # facet form: libm pow() per element, which is what `z[i] ** 2` is for the numpy scalar of
                # the 1-D path (and of the reference); numpy squares ARRAYS as x*x, one ulp off now and then
                # (as numpy scalars, not math.pow: a runaway profile overflows to inf with a warning, as in the 1-D path,
                # instead of raising)
                with np.errstate(over="ignore"):
                    return c.G * planet.mass / np.array([np.float64(v) ** 2 for v in zi])
            return c.G * planet.mass / zi ** 2

        below = np.unique(np.where(plevel > p_reference)[0])
        for i in below - 1:                                   # inwards from the reference level
            gravity[i] = g_at(i)
            dz[i] = c.k_b * tlevel[i] / (mmw[i] * gravity[i]) * np.log(plevel[i + 1] / plevel[i])
            z[i + 1] = z[i] - dz[i]
        for i in np.unique(np.where(plevel <= p_reference)[0])[::-1][:-1]:     # outwards
            gravity[i] = g_at(i)
            dz[i] = c.k_b * tlevel[i] / (mmw[i] * gravity[i]) * np.log(plevel[i] / plevel[i - 1])
            z[i - 1] = z[i] + dz[i]
        return self._finish_altitude(z, dz, gravity, g_at, tlevel, mmw)

    def _finish_altitude(self, z, dz, gravity, g_at, tlevel, mmw):
        dz[0] = dz[1]
        dz[-1] = dz[-2]
        self.level["z"], self.level["dz"] = z, dz
        self.layer["gravity"] = 0.5 * (gravity[:-1] + gravity[1:])
        gravity[-1], gravity[0] = g_at(-1), g_at(0)
        self.level["scale_height"] = self.c.k_b * tlevel / (mmw * gravity)

    def get_column_density(self):
        self.layer["colden"] = (self.level["pressure"][1:] - self.level["pressure"][:-1]) / self.layer["gravity"]

    def get_needed_continuum(self, available_ray_mol, available_continuum):
        """Continuum pairs and Rayleigh species present in both the profile and the opacity data
        (reference atmsetup.py:248-283; isotopologue name simplification is not needed here)."""
        names = list(self.molecules)
        self.continuum_molecules = []
        for m1 in names:
            for m2 in names:
                if m1 + m2 in available_continuum:
                    self.continuum_molecules += [[m1, m2]]
        if "H-" in names and "H-bf" in available_continuum:
            self.continuum_molecules += [["H-", "bf"]]
        if "H" in names and "electrons" in self.level.keys() and "H-ff" in available_continuum:
            self.continuum_molecules += [["H-", "ff"]]
        if "H2" in names and "electrons" in self.level.keys() and "H2-" in available_continuum:
            self.continuum_molecules += [["H2-", ""]]
        self.rayleigh_molecules = [m for m in names if m in available_ray_mol]

    def get_clouds(self, wno):
        nwno = np.size(wno)
        prof = self.input["clouds"]["profile"]
        self.cloud_free = prof is None
        if prof is None:       # (nlayer, nwno) zeros as read-only broadcast views: no 3 x 72 MB host arrays
            z = np.broadcast_to(np.zeros((self.c.nlayer, 1)), (self.c.nlayer, nwno))
            self.layer["cloud"] = {"w0": z, "g0": z, "opd": z}
            return
        in_wno = self.input["clouds"]["wavenumber"]
        sizes = {np.size(prof[k]) for k in ("opd", "g0", "w0")}
        if in_wno is not None and len(sizes) == 1:          # all three on their own grid: regridded where they are read
            nin = sizes.pop() // self.c.nlayer
            if nin == np.size(in_wno) and nin >= 2 and nin * self.c.nlayer == np.size(prof["opd"]) \
                    and not (nin == nwno and np.array_equal(in_wno, wno)) \
                    and bool(np.all(np.diff(np.asarray(in_wno, dtype=np.float64)) > 0)):   # else: numpy's own answer
                self.layer["cloud"] = CloudTables(
                    {k: np.ascontiguousarray(np.asarray(prof[k], dtype=np.float64).reshape(self.c.nlayer, nin))
                     for k in ("opd", "g0", "w0")}, in_wno, wno)
                return
        cld = {}
        for k in ("opd", "g0", "w0"):
            v = np.asarray(prof[k], dtype=np.float64)
            if v.ndim == 0:
                v = np.zeros((self.c.nlayer, nwno)) + v
            elif v.size == self.c.nlayer:
                v = np.repeat(v.reshape(self.c.nlayer, 1), nwno, axis=1)
            else:
                nin = v.size // self.c.nlayer
                v = v.reshape(self.c.nlayer, nin)
                if nin != nwno:
                    if in_wno is None:
                        raise Exception("cloud table has %d wavelengths, opacities %d: give "
                                        "clouds(wavenumber=...) to regrid" % (nin, nwno))
                    v = np.stack([np.interp(wno, in_wno, row) for row in v])   # wavelength.regrid
            cld[k] = np.ascontiguousarray(v)
        self.layer["cloud"] = cld

    def as_dict(self):
        """Picklable ``full_output`` dictionary with the reference's key names
        (reference atmsetup.py:704-790)."""
        bars = self.c.pconv
        out = {"weights": getattr(self, "weights", None),
               "layer": {"pressure_unit": "bars", "mixingratio_unit": "volume/volume",
                         "temperature_unit": "K", "pressure": self.layer["pressure"] / bars,
                         "mixingratios": self.layer["mixingratios"],
                         "temperature": self.layer["temperature"],
                         "column_density": self.layer.get("colden"), "mmw": self.layer.get("mmw"),
                         "cloud": {k: self.layer["cloud"][k] for k in ("w0", "g0", "opd")}},
               "wavenumber": getattr(self, "wavenumber", None), "wavenumber_unit": "cm-1",
               "level": {"pressure": self.level["pressure"] / bars,
                         "temperature": self.level["temperature"]},
               "latitude": getattr(self, "latitude", None), "longitude": getattr(self, "longitude", None),
               "star": {"flux_unit": "erg/cm2/s/cm"}, "warnings": self.warnings}
        for k in ("taugas", "tauray", "taucld"):
            out[k] = getattr(self, k, None)
        if getattr(self, "get_lvl_flux", False):
            for key, attr in (("thermal_fluxes", "lvl_output_thermal"), ("reflected_fluxes", "lvl_output_reflected")):
                if getattr(self, attr, None) is not None:
                    out["level"][key] = getattr(self, attr)
        for k in ("dz", "z"):
            if k in self.level:
                out["level"][k] = self.level[k]
        if hasattr(self, "xint_at_top"):
            out["albedo_3d"] = self.xint_at_top
            out["reflected_unit"] = "albedo"
        if hasattr(self, "flux_layers"):       # SH layer moment fluxes (calculate_fluxes='on')
            out["flux_layers"] = self.flux_layers
        if hasattr(self, "flux_at_top"):
            out["thermal_3d"] = self.flux_at_top
            out["thermal_unit"] = "erg/cm2/s/cm"
        return out
