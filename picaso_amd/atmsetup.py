"""Minimal atmosphere state for the accelerated path (counterpart of the reference ``ATMSETUP``,
picaso/atmsetup.py:17-876, restricted to what ``picaso()`` needs to feed the opacity and solver
kernels).  Host-side numpy only; everything is cgs (no astropy units).

Kept from the reference: level -> layer averaging (atmsetup.py:219-229), mean molecular weight
(:345-361), constant-gravity column density (:549-556), the cloud-free default and the
``(nlayer, nwno)`` reshape of cloud tables (:558-627), ``get_needed_continuum`` (:248-283).
Out of scope: altitude integration, chemistry, 3-D regridding, virga clouds.
"""
import numpy as np

# atomic masses (g/mol) for the species the synthetic / test configurations use
_ATOMIC = {"H": 1.00794, "He": 4.002602, "C": 12.0107, "N": 14.0067, "O": 15.9994, "Na": 22.98977,
           "K": 39.0983, "S": 32.065, "P": 30.97376, "Ti": 47.867, "V": 50.9415, "Fe": 55.845,
           "Si": 28.0855, "Mg": 24.305, "Al": 26.98154, "Ca": 40.078, "Cr": 51.9961, "Li": 6.941,
           "Rb": 85.4678, "Cs": 132.90545, "Cl": 35.453, "F": 18.9984}


def molecular_weight(name):
    """Molecular weight from a formula such as 'H2O', 'CH4', 'TiO' (case-sensitive elements)."""
    import re
    if name in ("e-", "H-"):
        return _ATOMIC["H"] if name == "H-" else 5.4858e-4
    name = name.rstrip("+-")
    total, pos = 0.0, 0
    for m in re.finditer(r"([A-Z][a-z]?)(\d*)", name):
        if m.start() != pos or m.group(1) not in _ATOMIC:
            raise KeyError(name)
        total += _ATOMIC[m.group(1)] * (int(m.group(2)) if m.group(2) else 1)
        pos = m.end()
    if pos != len(name) or total == 0:
        raise KeyError(name)
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
        w = np.zeros(self.c.nlevel)
        for m in self.molecules:
            w = w + self.level["mixingratios"][m] * self.weights[m]
        self.level["mmw"] = w
        self.layer["mmw"] = 0.5 * (w[:-1] + w[1:])

    def get_density(self):
        self.level["den"] = self.level["pressure"] / (self.c.k_b * self.level["temperature"])

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
        z = np.zeros(np.shape(tlevel)) + planet.radius
        dz = np.zeros(np.shape(tlevel))
        gravity = np.zeros(np.shape(tlevel))

        def g_at(i):
            return planet.gravity if constant_gravity else c.G * planet.mass / z[i] ** 2

        below = np.unique(np.where(plevel > p_reference)[0])
        for i in below - 1:                                   # inwards from the reference level
            gravity[i] = g_at(i)
            dz[i] = c.k_b * tlevel[i] / (mmw[i] * gravity[i]) * np.log(plevel[i + 1] / plevel[i])
            z[i + 1] = z[i] - dz[i]
        for i in np.unique(np.where(plevel <= p_reference)[0])[::-1][:-1]:     # outwards
            gravity[i] = g_at(i)
            dz[i] = c.k_b * tlevel[i] / (mmw[i] * gravity[i]) * np.log(plevel[i] / plevel[i - 1])
            z[i - 1] = z[i] + dz[i]
        dz[0] = dz[1]
        dz[-1] = dz[-2]
        self.level["z"], self.level["dz"] = z, dz
        self.layer["gravity"] = 0.5 * (gravity[:-1] + gravity[1:])
        gravity[-1], gravity[0] = g_at(-1), g_at(0)
        self.level["scale_height"] = c.k_b * tlevel / (mmw * gravity)

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
