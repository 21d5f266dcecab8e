"""Generate golden input/output vectors by running the REFERENCE's own source.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference modules (picaso/fluxes.py, picaso/disco.py) are imported under the identity-numba
shim of tools/ref_shim.py, called on seeded synthetic scenes (picaso_amd/synthetic.py) and the
inputs + outputs are written as small .npz fixtures next to this script.  A fixture is data only:
no reference source is stored.  Tests compare oracle/ (CPU) and the HIP library (GPU) against them.
"""
import itertools
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import ref_shim  # noqa: E402
from picaso_amd import synthetic as syn  # noqa: E402

fl = ref_shim.load("fluxes")
di = ref_shim.load("disco")

PLANES = ("dtau", "tau", "w0", "cosb", "gcos2", "ftau_cld", "ftau_ray", "dtau_og", "tau_og",
          "w0_og", "cosb_og")
TTHG = dict(frac_a=1.0, frac_b=-1.0, frac_c=2.0, constant_back=-0.5, constant_forward=1.0)


def geometry_1d(ng=5):
    g, gw, t, tw = di.get_angles_1d(ng)
    u0, u1, ct, lat, lon = di.compute_disco(ng, 1, g, t, 0.0)
    return dict(gangle=g, gweight=gw, tangle=t, tweight=tw, ubar0=u0, ubar1=u1, cos_theta=1.0,
                numg=ng, numt=1)


def geometry_3d(ng, nt, phase):
    g, gw, t, tw = di.get_angles_3d(ng, nt)
    u0, u1, ct, lat, lon = di.compute_disco(ng, nt, g, t, phase)
    return dict(gangle=g, gweight=gw, tangle=t, tweight=tw, ubar0=u0, ubar1=u1, cos_theta=ct,
                numg=ng, numt=nt)


def scenes_1d():
    """name -> (scene dict, geometry, surf_reflect, F0PI)"""
    out = {}
    nw = 16
    sc = syn.make_scene(90, nw, seed=3)
    out["cfg3like"] = (sc, geometry_1d(5), 0.0, np.ones(nw))
    sc = syn.make_scene(60, nw, seed=1, tkind="jupiter")
    out["jupiterlike"] = (sc, geometry_1d(6), 0.3, np.linspace(0.5, 2.0, nw))
    sc = syn.make_scene(12, nw, seed=5, gas_scale=1e-3, cloud_opd=0.02)
    out["thin"] = (sc, geometry_1d(5), np.linspace(0.0, 0.6, nw), np.ones(nw))
    sc = syn.make_scene(40, nw, seed=6, gas_scale=50.0, cloud_opd=40.0)
    out["thick"] = (sc, geometry_1d(8), 0.1, np.ones(nw))
    # conservative scattering, no delta-eddington, constant planes (test_mode style)
    cs = syn.delta_scale(syn.constant_scene(30, nw, 0.5, 0.999999, 0.5), delta_eddington=False)
    cs.update(wno=syn.wavenumber_grid(nw), nlayer=30, nlevel=31, nwno=nw)
    p, t = syn.pressure_temperature(31)
    cs.update(plevel=p * 1e6, tlevel=t)
    out["conservative"] = (cs, geometry_1d(5), 0.0, np.ones(nw))
    # 1-D planes seen under a non-zero phase angle: ng x nt angles share the planes
    sc = syn.make_scene(25, nw, seed=8)
    out["phase60"] = (sc, geometry_3d(4, 3, np.pi / 3), 0.2, np.ones(nw))
    sc = syn.make_scene(1, nw, seed=9, cloud=False)
    out["onelayer"] = (sc, geometry_1d(5), 0.25, np.ones(nw))
    sc = syn.make_scene(2, nw, seed=10, cloud=False)
    out["twolayer"] = (sc, geometry_1d(7), 0.0, np.ones(nw))
    return out


def run_reflected_1d(name, sc, geo, rs, f0, store):
    nlevel, nwno = sc["nlevel"], sc["nwno"]
    combos = list(itertools.product(range(4), range(2), range(2)))
    for sp, mp, tc in combos:
        for lvl in (0, 1):
            if lvl and (sp, mp, tc) not in ((3, 0, 0), (1, 1, 1)):
                continue
            args = (nlevel, sc["wno"], nwno, geo["numg"], geo["numt"]) + tuple(
                sc[k].copy() for k in PLANES) + (rs, geo["ubar0"], geo["ubar1"], geo["cos_theta"],
                                                 f0, sp, mp, TTHG["frac_a"], TTHG["frac_b"],
                                                 TTHG["frac_c"], TTHG["constant_back"],
                                                 TTHG["constant_forward"])
            b_top = 0.0 if name != "thin" else 0.01
            xint, lv = fl.get_reflected_1d(*args, get_toa_intensity=1, get_lvl_flux=lvl,
                                           toon_coefficients=tc, b_top=b_top)
            key = "refl1d/sp%d_mp%d_tc%d_lvl%d" % (sp, mp, tc, lvl)
            store[key + "/xint"] = xint
            store[key + "/b_top"] = np.array(b_top)
            if lvl:
                for nm, arr in zip(("fm", "fp", "fmm", "fpm"), lv):
                    store[key + "/" + nm] = arr
            if lvl == 0 and (sp, mp, tc) == (3, 0, 0):
                store["compress_disco/albedo"] = di.compress_disco(
                    nwno, geo["cos_theta"], xint, geo["gweight"], geo["tweight"], f0)


def _extended(fn, *args, **kw):
    """Evaluate a reference function in x87 extended precision (np.longdouble): every float64 array
    argument is widened and the module-level ``zeros`` the reference allocates its work arrays with
    is made to allocate longdouble.  Used to record how well-conditioned the reference's OWN fp64
    level fluxes are (deep, optically thick levels cancel catastrophically in the bottom boundary
    row), so that the parity tests can tell implementation error from the reference's rounding."""
    L = np.longdouble
    wide = [a.astype(L) if isinstance(a, np.ndarray) and a.dtype == np.float64 else a for a in args]
    orig = fl.zeros
    fl.zeros = lambda *a, **k: np.zeros(*a, dtype=L, **k)
    try:
        out = fn(*wide, **kw)
    finally:
        fl.zeros = orig
    return out


def run_thermal_1d(name, sc, geo, rs, store):
    nlevel, nwno = sc["nlevel"], sc["nwno"]
    dwno = np.gradient(sc["wno"])
    store["dwno"] = dwno
    for hs, ct in itertools.product((0, 1), (0, 1)):
        rsv = np.zeros(nwno) + rs
        flux, lv = fl.get_thermal_1d(nlevel, sc["wno"], nwno, geo["numg"], geo["numt"],
                                     sc["tlevel"], sc["dtau_og"].copy(), sc["w0_no_raman"].copy(),
                                     sc["cosb_og"].copy(), sc["plevel"], geo["ubar1"], rsv, hs,
                                     dwno, ct)
        key = "therm1d/hs%d_ct%d" % (hs, ct)
        store[key + "/flux"] = flux
        if hs == ct:
            for nm, arr in zip(("fm", "fp", "fmm", "fpm"), lv):
                store[key + "/" + nm] = arr
            _, lvx = _extended(fl.get_thermal_1d, nlevel, sc["wno"], nwno, geo["numg"],
                               geo["numt"], sc["tlevel"], sc["dtau_og"].copy(),
                               sc["w0_no_raman"].copy(), sc["cosb_og"].copy(), sc["plevel"],
                               geo["ubar1"], rsv, hs, dwno, ct)
            for nm, arr in zip(("fm", "fp", "fmm", "fpm"), lvx):
                store[key + "/" + nm + "_x80"] = np.asarray(arr, dtype=np.float64)
        if (hs, ct) == (0, 0):
            store["compress_thermal/flux"] = di.compress_thermal(nwno, flux, geo["gweight"],
                                                                 geo["tweight"])
            store["compress_thermal/lvl_fp"] = di.compress_thermal(nwno, lv[1], geo["gweight"],
                                                                   geo["tweight"])


def make_1d():
    for name, (sc, geo, rs, f0) in scenes_1d().items():
        store = {}
        for k in PLANES + ("wno", "tlevel", "plevel", "w0_no_raman"):
            store["in/" + k] = sc[k]
        if "f_deltaM" in sc:
            store["in/f_deltaM"] = sc["f_deltaM"]
        store["in/surf_reflect"] = np.asarray(rs, dtype=float)
        store["in/F0PI"] = f0
        for k, v in geo.items():
            store["geo/" + k] = np.asarray(v)
        for k, v in TTHG.items():
            store["opt/" + k] = np.array(v)
        run_reflected_1d(name, sc, geo, rs, f0, store)
        run_thermal_1d(name, sc, geo, rs, store)
        path = os.path.join(HERE, "scene1d_%s.npz" % name)
        np.savez_compressed(path, **store)
        print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def make_3d():
    """get_reflected_3d / get_thermal_3d: per-facet planes (nlayer,nwno,ng,nt)."""
    for name, ng, nt, phase, nlayer in (("f3x3", 3, 3, np.pi / 3, 20), ("f4x2", 4, 2, 2.2, 33)):
        nw = 12
        geo = geometry_3d(ng, nt, phase)
        rng = np.random.default_rng(77 + ng)
        facets = [[syn.make_scene(nlayer, nw, seed=100 + 10 * g + t,
                                  cloud_opd=float(rng.uniform(0.05, 3.0)))
                   for t in range(nt)] for g in range(ng)]
        store = {}
        planes3 = {}
        for k in PLANES + ("w0_no_raman",):
            arr = np.zeros(facets[0][0][k].shape + (ng, nt))
            for g in range(ng):
                for t in range(nt):
                    arr[:, :, g, t] = facets[g][t][k]
            planes3[k] = arr
            store["in/" + k] = arr
        p, tl = syn.pressure_temperature(nlayer + 1)
        t3 = np.zeros((nlayer + 1, ng, nt))
        p3 = np.zeros((nlayer + 1, ng, nt))
        for g in range(ng):
            for t in range(nt):
                t3[:, g, t] = tl * (1.0 + 0.1 * rng.uniform(-1, 1))
                p3[:, g, t] = p * 1e6
        wno = syn.wavenumber_grid(nw)
        f0 = np.linspace(0.8, 1.3, nw)
        rs = np.linspace(0.0, 0.4, nw)
        store.update({"in/tlevel": t3, "in/plevel": p3, "in/wno": wno, "in/F0PI": f0,
                      "in/surf_reflect": rs})
        for k, v in geo.items():
            store["geo/" + k] = np.asarray(v)
        for k, v in TTHG.items():
            store["opt/" + k] = np.array(v)
        for sp, mp in itertools.product(range(4), range(2)):
            xint = fl.get_reflected_3d(nlayer + 1, wno, nw, ng, nt,
                                       *[planes3[k].copy() for k in PLANES], rs, geo["ubar0"],
                                       geo["ubar1"], geo["cos_theta"], f0, sp, mp,
                                       TTHG["frac_a"], TTHG["frac_b"], TTHG["frac_c"],
                                       TTHG["constant_back"], TTHG["constant_forward"])
            store["refl3d/sp%d_mp%d/xint" % (sp, mp)] = xint
            if (sp, mp) == (0, 0):
                store["compress_disco/albedo"] = di.compress_disco(
                    nw, geo["cos_theta"], xint, geo["gweight"], geo["tweight"], f0)
        for hs in (0, 1):
            flux = fl.get_thermal_3d(nlayer + 1, wno, nw, ng, nt, t3, planes3["dtau_og"].copy(),
                                     planes3["w0_no_raman"].copy(), planes3["cosb_og"].copy(), p3,
                                     geo["ubar1"], rs, hs)
            store["therm3d/hs%d/flux" % hs] = flux
            if hs == 0:
                store["compress_thermal/flux"] = di.compress_thermal(nw, flux, geo["gweight"],
                                                                     geo["tweight"])
        path = os.path.join(HERE, "scene3d_%s.npz" % name)
        np.savez_compressed(path, **store)
        print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def make_geometry():
    store = {}
    for n in (5, 6, 7, 8):
        g, gw, t, tw = di.get_angles_1d(n)
        store["angles1d/%d/gangle" % n] = g
        store["angles1d/%d/gweight" % n] = gw
    for ng, nt, ph in ((5, 1, 0.0), (6, 4, 1.0), (8, 8, np.pi / 3), (10, 10, 4.0), (3, 3, 3.0)):
        if nt == 1:
            g, gw, t, tw = di.get_angles_1d(ng)
        else:
            g, gw, t, tw = di.get_angles_3d(ng, nt)
        u0, u1, ct, lat, lon = di.compute_disco(ng, nt, g, t, ph)
        key = "disco/%dx%d_%.4f" % (ng, nt, ph)
        for nm, v in zip(("gangle", "gweight", "tangle", "tweight", "ubar0", "ubar1", "cos_theta",
                          "lat", "lon"), (g, gw, t, tw, u0, u1, ct, lat, lon)):
            store[key + "/" + nm] = np.asarray(v)
    path = os.path.join(HERE, "geometry.npz")
    np.savez_compressed(path, **store)
    print("wrote", path)


if __name__ == "__main__":
    which = sys.argv[1:] or ["1d", "3d", "geometry"]
    if "1d" in which:
        make_1d()
    if "3d" in which:
        make_3d()
    if "geometry" in which:
        make_geometry()


# ------------------------------------------------------------------------------------------------
# opacity pre-stage: RetrieveOpacities.get_opacities[_nearest] + compute_opacity
# ------------------------------------------------------------------------------------------------
def _ref_colden(p_dyn, tlevel, mmw_lvl, gravity, p_reference=1):
    """Layer column density exactly as the reference's ATMSETUP produces it for a gravity-only planet
    (get_altitude + get_column_density, atmsetup.py:384-461, :549-555; justdoit.py:204-206)."""
    am = ref_shim.load("atmsetup")

    class Bare:
        pass
    s = Bare()
    s.c, s.planet, s.level, s.layer = Bare(), Bare(), {}, {}
    s.c.pconv, s.c.k_b, s.c.amu, s.c.G, s.c.nlevel = 1e6, 1.380649e-16, 1.66053906660e-24, 6.6743e-8, len(p_dyn)
    s.planet.radius, s.planet.mass, s.planet.gravity = np.nan, np.nan, gravity
    s.level.update(mmw=mmw_lvl, temperature=tlevel, pressure=p_dyn)
    am.ATMSETUP.get_altitude(s, p_reference=p_reference)
    am.ATMSETUP.get_column_density(s)
    return s.layer["colden"]


def _ref_weights(names):
    """Molecular weights from the reference's ATMSETUP.get_weights (main-isotope masses)."""
    am = ref_shim.load("atmsetup")
    return {k: float(am.ATMSETUP.get_weights(object(), [k])[k]) for k in names}


def make_optics(nwno=40, nlevel=31, tag="", full=True, qms=None):
    """Synthetic monochromatic sqlite DB in the reference schema (committed next to the fixtures),
    driven through the reference's own RetrieveOpacities + compute_opacity with a duck-typed
    atmosphere.  Defaults: the 40-point x 30-layer fixture of round 1;
    ``make_optics(196, 61, "_196x60", full=False)`` is BASELINE configs[0]'s shape (196-point opacity
    grid, 60 layers) with the planes of the default options only (``qms``: the query methods, default linear only;
    round 6 adds "nearest" AFTER "linear", so the arrays of rounds 1-5 are reproduced unchanged)."""
    import sqlite3
    import types
    import pandas as pd
    optics = ref_shim.load("optics")
    rayleigh = ref_shim.load("rayleigh")
    rng = np.random.default_rng(4242)
    wno = np.linspace(4000.0, 25000.0, nwno)
    temps = [100.0, 300.0, 700.0, 1500.0, 3000.0]
    press = [1e-6, 1e-4, 1e-2, 1.0, 100.0, 500.0]
    mols = ["H2O", "CH4", "H2"]
    cont_pairs = ["H2H2", "H2He", "H2CH4"]
    cia_temps = [75.0, 200.0, 500.0, 1000.0, 2000.0, 4000.0]
    db = os.path.join(HERE, "synthetic_opacities%s.db" % tag)
    if os.path.exists(db):
        os.remove(db)

    def adapt(arr):
        out = io.BytesIO()
        np.save(out, arr)
        out.seek(0)
        return sqlite3.Binary(out.read())
    import io
    sqlite3.register_adapter(np.ndarray, adapt)
    sqlite3.register_adapter(np.int64, int)
    conn = sqlite3.connect(db)
    cur = conn.cursor()
    cur.execute("CREATE TABLE header (id INTEGER PRIMARY KEY, pressure_unit VARCHAR, temperature_unit "
                "VARCHAR, wavenumber_grid array, continuum_unit VARCHAR, molecular_unit VARCHAR)")
    cur.execute("CREATE TABLE molecular (id INTEGER PRIMARY KEY, ptid INTEGER, molecule VARCHAR, "
                "pressure FLOAT, temperature FLOAT, opacity array)")
    cur.execute("CREATE TABLE continuum (id INTEGER PRIMARY KEY, molecule VARCHAR, temperature FLOAT, "
                "opacity array)")
    cur.execute("CREATE TABLE rayleigh (id INTEGER PRIMARY KEY, molecule VARCHAR, opacity array)")
    cur.execute("INSERT INTO header (pressure_unit, temperature_unit, wavenumber_grid, continuum_unit, "
                "molecular_unit) VALUES (?,?,?,?,?)", ("bar", "kelvin", wno, "cm-1 amagat-2", "cm2/molecule"))
    ptid = 0
    for t in temps:
        for p in press:
            ptid += 1
            for m in mols:
                lg = (-24.0 + 2.0 * np.sin(wno / 2500.0 + mols.index(m)) + 0.4 * np.log10(p)
                      + 0.8 * np.log10(t / 300.0) + 0.2 * rng.standard_normal(nwno))
                k = 10.0 ** lg
                if m == "CH4":
                    k[::7] = 0.0            # exercise the zero -> 1e-50 floor (optics.py:2282)
                cur.execute("INSERT INTO molecular (ptid, molecule, pressure, temperature, opacity) "
                            "VALUES (?,?,?,?,?)", (ptid, m, p, t, k))
    for pair in cont_pairs:
        for t in cia_temps:
            k = 10.0 ** (-7.0 + np.cos(wno / 4000.0 + cont_pairs.index(pair)) + 0.3 * np.log10(t / 300.0))
            cur.execute("INSERT INTO continuum (molecule, temperature, opacity) VALUES (?,?,?)",
                        (pair, t, k))
    ray = rayleigh.Rayleigh(wno)
    ray_opa = {m: ray.compute_sigma(m) for m in ("H2", "He", "CH4", "H2O")}
    for m, v in ray_opa.items():
        cur.execute("INSERT INTO rayleigh (molecule, opacity) VALUES (?,?)", (m, v))
    conn.commit()
    conn.close()

    nlayer = nlevel - 1
    plevel_bar = np.logspace(-5.5, 1.8, nlevel)
    tlevel = 150.0 + 1200.0 * ((np.log10(plevel_bar) + 5.5) / 7.3) ** 2
    mix = {"H2": np.full(nlevel, 0.84), "He": np.full(nlevel, 0.155),
           "H2O": np.linspace(1e-4, 3e-3, nlevel), "CH4": np.linspace(4e-4, 1e-3, nlevel)}
    gravity = 2500.0
    weights = _ref_weights(("H2", "He", "H2O", "CH4"))
    c0, c1 = (14 * nlayer) // 30, (20 * nlayer) // 30         # cloud slab (layers 14..19 of 30)
    cld_opd = np.zeros((nlayer, nwno))
    cld_opd[c0:c1] = 0.3 * (1.0 + 0.2 * np.sin(wno / 3000.0))
    cld_w0 = np.zeros((nlayer, nwno))
    cld_w0[c0:c1] = 0.93
    cld_g0 = np.zeros((nlayer, nwno))
    cld_g0[c0:c1] = 0.65
    shifts_n = None

    def make_atm():
        atm = types.SimpleNamespace()
        c = types.SimpleNamespace(nlayer=nlayer, nlevel=nlevel, pconv=1e6, k_b=1.380649e-16,
                                  amu=1.66053906660e-24, rgas=8.31446261815324)
        atm.c = c
        atm.planet = types.SimpleNamespace(gravity=gravity)
        p = plevel_bar * 1e6
        atm.level = {"pressure": p, "temperature": tlevel}
        lay_mix = pd.DataFrame({k: 0.5 * (v[1:] + v[:-1]) for k, v in mix.items()})
        mmw_lvl = sum(mix[k] * weights[k] for k in mix)
        atm.layer = {"pressure": np.sqrt(p[1:] * p[:-1]), "temperature": 0.5 * (tlevel[1:] + tlevel[:-1]),
                     "mmw": 0.5 * (mmw_lvl[1:] + mmw_lvl[:-1]), "colden": _ref_colden(p, tlevel, mmw_lvl, gravity),
                     "electrons": np.zeros(nlayer), "mixingratios": lay_mix,
                     "cloud": {"opd": cld_opd.copy(), "w0": cld_w0.copy(), "g0": cld_g0.copy()}}
        atm.molecules = np.array(["H2O", "CH4", "H2"])
        atm.continuum_molecules = [["H2", "H2"], ["H2", "He"], ["H2", "CH4"]]
        atm.rayleigh_molecules = ["H2", "He", "CH4", "H2O"]
        return atm

    store = {"in/plevel_bar": plevel_bar, "in/tlevel": tlevel, "in/gravity": np.array(gravity),
             "in/colden": make_atm().layer["colden"],
             "in/cld_opd": cld_opd, "in/cld_w0": cld_w0, "in/cld_g0": cld_g0, "in/wno": wno}
    for k, v in mix.items():
        store["in/mix/" + k] = v
        store["in/weight/" + k] = np.array(weights[k])
    names = ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray", "gcos2", "dtau_og", "tau_og",
             "w0_og", "cosb_og", "w0_no_raman", "f_deltaM")
    raman_file = os.path.join(ref_shim.REF_ROOT, "reference", "opacities", "raman.txt")
    for qm in (qms or (("nearest", "linear") if full else ("linear",))):
        opa = optics.RetrieveOpacities(db, raman_file, query_method=qm)
        if shifts_n is None:
            shifts_n = 1.0 + 0.05 * rng.standard_normal((nwno, len(opa.raman_db)))
            store["in/raman_shifts"] = shifts_n
            store["in/raman_c"] = opa.raman_db["c"].values
            store["in/raman_ji"] = opa.raman_db["ji"].values
            store["in/raman_deltanu"] = opa.raman_db["deltanu"].values
        opa.raman_stellar_shifts = shifts_n
        atm = make_atm()
        opa.get_opacities(atm)
        for m in atm.molecules:
            store["%s/molecular_opa/%s" % (qm, m)] = opa.molecular_opa[m]
        for pr in ("H2H2", "H2He", "H2CH4"):
            store["%s/continuum_opa/%s" % (qm, pr)] = opa.continuum_opa[pr]
        store["%s/pt_opa_index" % qm] = np.asarray(atm.layer["pt_opa_index"])
        for de, stream, raman, tm in (((True, 2, 2, None), (False, 2, 2, None), (True, 4, 0, None),
                                       (True, 2, 2, "rayleigh"), (False, 2, 2, "constant_tau"),
                                       (True, 2, 1, None)) if full else ((True, 2, 2, None),)):
            atm = make_atm()
            opa.get_opacities(atm)
            out = optics.compute_opacity(atm, opa, ngauss=1, stream=stream, delta_eddington=de,
                                         test_mode=tm, raman=raman)
            key = "%s/de%d_s%d_r%d_tm%s" % (qm, int(de), stream, raman, tm or "none")
            for nm, arr in zip(names, out):
                store[key + "/" + nm] = np.asarray(arr)[:, :, 0]
        if not full:
            continue
        # patchy clouds: the thinned-cloud column set (optics.py:314-315, justdoit.py:248-252)
        atm = make_atm()
        opa.get_opacities(atm)
        out = optics.compute_opacity(atm, opa, ngauss=1, stream=2, delta_eddington=True, test_mode=None, raman=2,
                                     fthin_cld=0.1, do_holes=True)
        for nm, arr in zip(names, out):
            store["%s/holes_fthin0.1/%s" % (qm, nm)] = np.asarray(arr)[:, :, 0]
    path = os.path.join(HERE, "optics%s.npz" % tag)
    np.savez_compressed(path, **store)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), "and", db,
          "%.1f KB" % (os.path.getsize(db) / 1024))


def make_ck():
    """Premixed correlated-k path: a synthetic ln(kappa) table (ragged pressure grid per temperature,
    4 Gauss points) driven through the reference's own RetrieveCKs.get_pre_mix_ck / get_continuum
    (bound to a bare instance: the constructor only parses file formats that are not available
    here) and compute_opacity(ngauss=4); continuum rows come from synthetic_opacities.db."""
    import types
    import pandas as pd
    optics = ref_shim.load("optics")
    db = os.path.join(HERE, "synthetic_opacities.db")
    assert os.path.exists(db), "run `make_golden.py optics` first"
    rng = np.random.default_rng(777)
    og = np.load(os.path.join(HERE, "optics.npz"))
    wno = og["in/wno"]
    nwno = wno.size
    ngauss = 4
    gauss_wts = np.array([0.4, 0.3, 0.2, 0.1])
    temps = np.array([100.0, 250.0, 600.0, 1200.0, 2600.0])
    press = np.array([1e-6, 1e-4, 1e-2, 1.0, 30.0, 300.0])
    nc_p = np.array([6, 6, 6, 5, 4])                      # hot temperatures lack the highest pressures
    # the reference keeps one entry per (P,T) pair (optics.py:1556-1575 style lists)
    pressures = np.concatenate([press[:n] for n in nc_p])
    temps_flat = np.concatenate([[t] * n for t, n in zip(temps, nc_p)])
    kappa = np.zeros((press.size, temps.size, nwno, ngauss))
    for ip, p_ in enumerate(press):
        for it, t_ in enumerate(temps):
            base = (-26.0 + 2.0 * np.sin(wno / 2500.0) + 0.5 * np.log10(p_) + 0.9 * np.log10(t_ / 300.0))
            for ig in range(ngauss):
                kappa[ip, it, :, ig] = np.log(10.0) * (base + 0.9 * ig + 0.1 * rng.standard_normal(nwno))
    cia_temps = np.array([75.0, 200.0, 500.0, 1000.0, 2000.0, 4000.0])

    nlevel = 31
    nlayer = nlevel - 1
    plevel_bar = og["in/plevel_bar"]
    tlevel = og["in/tlevel"].copy()
    mixkeys = ("H2", "He", "H2O", "CH4")
    mix = {k: og["in/mix/" + k] for k in mixkeys}
    gravity = float(og["in/gravity"])
    weights = _ref_weights(("H2", "He", "H2O", "CH4"))

    def make_atm():
        atm = types.SimpleNamespace()
        atm.c = types.SimpleNamespace(nlayer=nlayer, nlevel=nlevel, pconv=1e6, k_b=1.380649e-16,
                                      amu=1.66053906660e-24, rgas=8.31446261815324)
        atm.planet = types.SimpleNamespace(gravity=gravity)
        p = plevel_bar * 1e6
        atm.level = {"pressure": p, "temperature": tlevel}
        lay_mix = pd.DataFrame({k: 0.5 * (v[1:] + v[:-1]) for k, v in mix.items()})
        mmw_lvl = sum(mix[k] * weights[k] for k in mix)
        atm.layer = {"pressure": np.sqrt(p[1:] * p[:-1]), "temperature": 0.5 * (tlevel[1:] + tlevel[:-1]),
                     "mmw": 0.5 * (mmw_lvl[1:] + mmw_lvl[:-1]), "colden": _ref_colden(p, tlevel, mmw_lvl, gravity),
                     "electrons": np.zeros(nlayer), "mixingratios": lay_mix,
                     "cloud": {"opd": og["in/cld_opd"].copy(), "w0": og["in/cld_w0"].copy(),
                               "g0": og["in/cld_g0"].copy()}}
        atm.molecules = np.array(["H2O", "CH4", "H2"])
        atm.continuum_molecules = [["H2", "H2"], ["H2", "He"], ["H2", "CH4"]]
        atm.rayleigh_molecules = ["H2", "He", "CH4", "H2O"]
        return atm

    opa = object.__new__(optics.RetrieveCKs)
    opa.pressures, opa.temps, opa.nc_p, opa.kappa = pressures, temps_flat, nc_p, kappa
    opa.continuum_db, opa.cia_temps = db, cia_temps
    opa.wno, opa.nwno, opa.ngauss, opa.gauss_wts = wno, nwno, ngauss, gauss_wts
    rayleigh = ref_shim.load("rayleigh")
    ray = rayleigh.Rayleigh(wno)
    opa.rayleigh_opa = {m: ray.compute_sigma(m) for m in ("H2", "He", "CH4", "H2O")}
    atm = make_atm()
    opa.get_pre_mix_ck(atm)
    opa.get_continuum(atm)
    store = {"in/press": press, "in/temps": temps, "in/nc_p": nc_p, "in/kappa": kappa,
             "in/gauss_wts": gauss_wts, "in/cia_temps": cia_temps, "molecular_opa": opa.molecular_opa}
    for pr in ("H2H2", "H2He", "H2CH4"):
        store["continuum_opa/" + pr] = opa.continuum_opa[pr]
    names = ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray", "gcos2", "dtau_og", "tau_og",
             "w0_og", "cosb_og", "w0_no_raman", "f_deltaM")
    for de, stream in ((True, 2), (False, 2), (True, 4)):
        atm = make_atm()
        opa.get_pre_mix_ck(atm)
        opa.get_continuum(atm)
        out = optics.compute_opacity(atm, opa, ngauss=ngauss, stream=stream, delta_eddington=de, raman=2,
                                     test_mode=None)
        for nm, arr in zip(names, out):
            store["de%d_s%d/%s" % (int(de), stream, nm)] = np.asarray(arr)
    # on-the-fly mixing of per-gas tables through the reference's own methods
    # (RetrieveCKs.get_mixing_indices / mix_my_opacities_gasesfly, optics.py:1164-1278)
    rng = np.random.default_rng(31)
    xg, wg = np.polynomial.legendre.leggauss(ngauss)
    opa.gauss_pts = list(0.5 * (xg + 1))
    opa.kappas = {}
    for m in ("H2O", "CH4", "H2"):
        base = kappa[:, :, :, :1] - 3.0 * rng.random((kappa.shape[0], kappa.shape[1], kappa.shape[2], 1))
        opa.kappas[m] = base + np.cumsum(rng.random(kappa.shape) * 1.5, axis=3)
        store["fly/kappas/" + m] = opa.kappas[m]
    store["fly/gauss_pts"] = np.array(opa.gauss_pts)
    opa.temps = temps            # the per-gas table loader keeps the unique temperatures (get_ck_tables)
    atm = make_atm()
    idx, t_i, p_i = opa.get_mixing_indices(atm)
    store["fly/indices"], store["fly/t_interp"], store["fly/p_interp"] = idx, t_i, p_i
    opa.mix_my_opacities_gasesfly(atm)
    store["fly/molecular_opa"] = opa.molecular_opa
    atm = make_atm()
    opa.mix_my_opacities_gasesfly(atm, exclude_mol={"H2O": 1, "CH4": 0, "H2": 1})
    store["fly/molecular_opa_noCH4"] = opa.molecular_opa
    atm = make_atm()
    opa.get_continuum(atm)
    opa.mix_my_opacities_gasesfly(atm)
    out = optics.compute_opacity(atm, opa, ngauss=ngauss, stream=2, delta_eddington=True, raman=2, test_mode=None)
    for nm, arr in zip(names, out):
        store["fly/de1_s2/" + nm] = np.asarray(arr)
    path = os.path.join(HERE, "ck.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))



def make_transit():
    """get_transit_1d of the reference on a hydrostatic isothermal-ish column (z decreasing from the
    top), called as justdoit.py:390-394 calls it: LEVEL pressure / temperature in the player / tlayer
    slots."""
    fluxes = ref_shim.load("fluxes")
    rng = np.random.default_rng(99)
    store = {}
    for name, nlevel, nwno in (("a", 31, 24), ("b", 61, 16), ("two", 3, 8)):
        nl = nlevel - 1
        k_b, amu, G = 1.380649e-16, 1.66053906660e-24, 6.67430e-8
        plevel = np.logspace(-6, 1.5, nlevel) * 1e6
        tlevel = 600.0 + 500.0 * np.linspace(0, 1, nlevel) ** 2
        mmw = 2.3 + 0.2 * rng.random(nl)
        radius, gravity = 7.0e9, 2500.0
        # hydrostatic altitude above the deepest level, decreasing with index
        H = k_b * 0.5 * (tlevel[1:] + tlevel[:-1]) / (mmw * amu * gravity)
        dzl = H * np.log(plevel[1:] / plevel[:-1])
        z = radius + np.concatenate([np.cumsum(dzl[::-1])[::-1], [0.0]])
        dz = np.concatenate([dzl, [dzl[-1]]])
        colden = (plevel[1:] - plevel[:-1]) / gravity
        wl = np.linspace(0, 1, nwno)
        dtau = (10 ** (-6 + 7 * np.linspace(0, 1, nl))[:, None]) * (1 + 5 * np.sin(6 * wl)[None, :] ** 2) \
            * (1 + 0.1 * rng.random((nl, nwno)))
        rstar = 6.96e10
        F = fluxes.get_transit_1d(z, dz, nlevel, nwno, rstar, mmw, k_b, amu, plevel, tlevel, colden, dtau)
        for k, v in dict(z=z, dz=dz, mmw=mmw, plevel=plevel, tlevel=tlevel, colden=colden, dtau=dtau,
                         rstar=np.array(rstar), k_b=np.array(k_b), amu=np.array(amu), F=F).items():
            store["%s/%s" % (name, k)] = v
    path = os.path.join(HERE, "transit.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def make_sh():
    """get_reflected_SH / get_thermal_SH (stream 2 and 4) on three of the 1-D scenes.  f_deltaM is
    handed over as a fresh copy each call (the reference compounds it in place per angle)."""
    sc_all = scenes_1d()
    for name in ("cfg3like", "phase60", "thick", "conservative"):
        sc, geo, rs, f0 = sc_all[name]
        nlevel, nwno = sc["nlevel"], sc["nwno"]
        store = {}
        for k in PLANES + ("wno", "tlevel", "plevel", "w0_no_raman", "f_deltaM"):
            store["in/" + k] = sc[k]
        store["in/surf_reflect"] = np.asarray(rs, dtype=float)
        store["in/F0PI"] = f0
        for k, v in geo.items():
            store["geo/" + k] = np.asarray(v)
        for k, v in TTHG.items():
            store["opt/" + k] = np.array(v)
        combos = [(0, 0, 0, 1, 1, 1, 0), (1, 1, 1, 1, 1, 1, 0), (0, 0, 0, 0, 0, 0, 0),
                  (1, 0, 0, 1, 1, 1, 1), (1, 1, 1, 0, 1, 0, 1)]
        for stream in (2, 4):
            for (wsf, wmf, psf, wsr, wmr, psr, sf) in combos:
                fd = sc["f_deltaM"].copy()
                if stream == 4:      # f_deltaM = cosb**stream for the delta-M scaled scenes
                    fd = sc["cosb_og"] ** 4 if np.any(sc["f_deltaM"]) else fd
                store["in/f_deltaM_s%d" % stream] = fd.copy()
                want_flux = (wsf, wmf, psf, wsr, wmr, psr, sf) == combos[0]
                xint, flux = fl.get_reflected_SH(
                    nlevel, nwno, geo["numg"], geo["numt"], sc["dtau"].copy(), sc["tau"].copy(),
                    sc["w0"].copy(), sc["cosb"].copy(), sc["ftau_cld"].copy(), sc["ftau_ray"].copy(),
                    fd, sc["dtau_og"].copy(), sc["tau_og"].copy(), sc["w0_og"].copy(),
                    sc["cosb_og"].copy(), rs, geo["ubar0"], geo["ubar1"], geo["cos_theta"], f0, wsf,
                    wmf, psf, wsr, wmr, psr, TTHG["frac_a"], TTHG["frac_b"], TTHG["frac_c"],
                    TTHG["constant_back"], TTHG["constant_forward"], stream, b_top=0.0,
                    flx=1 if want_flux else 0, single_form=sf)
                store["reflsh/s%d_f%d%d%d_r%d%d%d_sf%d/xint" % (stream, wsf, wmf, psf, wsr, wmr, psr, sf)] = xint
                if want_flux:       # layer moment fluxes F.X + G (calculate_flux, fluxes.py:3631-3635)
                    store["reflsh/s%d_f%d%d%d_r%d%d%d_sf%d/flux" % (stream, wsf, wmf, psf, wsr, wmr, psr, sf)] = flux
            for hs in (0, 1):
                rsv = np.zeros(nwno) + rs
                xint, _ = fl.get_thermal_SH(nlevel, sc["wno"], nwno, geo["numg"], geo["numt"],
                                            sc["tlevel"], sc["dtau"].copy(), sc["tau"].copy(),
                                            sc["w0"].copy(), sc["cosb"].copy(), sc["dtau_og"].copy(),
                                            sc["tau_og"].copy(), sc["w0_og"].copy(),
                                            sc["w0_no_raman"].copy(), sc["cosb_og"].copy(),
                                            sc["plevel"], geo["ubar1"], rsv, stream, hs)
                store["thermsh/s%d_hs%d/xint" % (stream, hs)] = xint
        path = os.path.join(HERE, "scene_sh_%s.npz" % name)
        np.savez_compressed(path, **store)
        print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__" and (("optics" in sys.argv[1:]) or not sys.argv[1:]):
    make_optics()
if __name__ == "__main__" and (("optics196" in sys.argv[1:]) or not sys.argv[1:]):
    make_optics(196, 61, "_196x60", full=False, qms=("linear", "nearest"))
if __name__ == "__main__" and (("ck" in sys.argv[1:]) or not sys.argv[1:]):
    make_ck()
if __name__ == "__main__" and (("transit" in sys.argv[1:]) or not sys.argv[1:]):
    make_transit()
if __name__ == "__main__" and (("sh" in sys.argv[1:]) or not sys.argv[1:]):
    make_sh()


def make_altitude():
    """ATMSETUP.get_altitude + get_column_density of the reference (atmsetup.py:384-461, :549-555),
    called unbound on a bare object carrying the fields they read (astropy is absent here, so the
    constants are given in cgs as astropy would)."""
    am = ref_shim.load("atmsetup")

    class Bare:
        pass
    store = {}
    cases = dict(
        const_g=dict(nlevel=12, radius=np.nan, mass=np.nan, gravity=2500.0, p_reference=1.0, plo=-5, phi=2),
        r_and_m=dict(nlevel=31, radius=7.1e9, mass=1.9e30, gravity=2516.0, p_reference=10.0, plo=-6, phi=2),
        pref_deep=dict(nlevel=9, radius=6.4e8, mass=6.0e27, gravity=978.0, p_reference=1e4, plo=-4, phi=1),
        pref_top=dict(nlevel=7, radius=2.5e9, mass=1.0e29, gravity=1068.0, p_reference=1e-9, plo=-3, phi=1),
        const_g_deep=dict(nlevel=10, radius=np.nan, mass=np.nan, gravity=1500.0, p_reference=1e4, plo=-4, phi=1),
        const_g_top=dict(nlevel=6, radius=np.nan, mass=np.nan, gravity=900.0, p_reference=1e-9, plo=-3, phi=1),
        const_g_two=dict(nlevel=2, radius=np.nan, mass=np.nan, gravity=900.0, p_reference=1.0, plo=-1, phi=1),
    )
    rng = np.random.default_rng(5)
    for name, cs in cases.items():
        n = cs["nlevel"]
        s = Bare()
        s.c, s.planet, s.level, s.layer = Bare(), Bare(), {}, {}
        s.c.pconv, s.c.k_b, s.c.amu, s.c.G, s.c.nlevel = 1e6, 1.380649e-16, 1.66053906660e-24, 6.6743e-8, n
        s.planet.radius, s.planet.mass, s.planet.gravity = cs["radius"], cs["mass"], cs["gravity"]
        s.level["mmw"] = 2.2 + 0.3 * rng.random(n)
        s.level["temperature"] = 150.0 + 1200.0 * np.linspace(0, 1, n) ** 1.5
        s.level["pressure"] = np.logspace(cs["plo"], cs["phi"], n) * 1e6
        am.ATMSETUP.get_altitude(s, p_reference=cs["p_reference"])
        am.ATMSETUP.get_column_density(s)
        for k in ("radius", "mass", "gravity", "p_reference"):
            store["%s/%s" % (name, k)] = np.array(cs[k])
        for k in ("mmw", "temperature", "pressure", "z", "dz", "scale_height"):
            store["%s/%s" % (name, k)] = s.level[k]
        store["%s/layer_gravity" % name] = s.layer["gravity"]
        store["%s/colden" % name] = s.layer["colden"]
    # molecular weights of the species PICASO's opacity databases and chemistry tables name
    mols = ["H2", "He", "H2O", "CH4", "CO", "CO2", "NH3", "N2", "Na", "K", "TiO", "VO", "FeH", "H2S", "PH3", "HCN",
            "C2H2", "C2H4", "C2H6", "O2", "O3", "SO2", "Fe", "H", "Li", "Rb", "Cs", "CrH", "MgH", "SiO", "OCS",
            "LiCl", "LiH", "LiF", "H-", "H+", "H3+", "NO", "NO2", "N2O", "CaH", "AlH", "OH", "CH3D", "HDO", "Ar",
            "Ne", "Kr", "Xe", "SiH4", "GeH4", "AsH3", "HCl", "HF", "NaH", "KCl", "Mg", "Ca", "Al", "Ti", "V", "Cr",
            "Mn", "Ni", "Zn", "C", "N", "O", "S", "P", "Si", "graphite"]
    wts = _ref_weights(mols)
    store["weights/names"] = np.array(mols)
    store["weights/values"] = np.array([wts[m] for m in mols])
    path = os.path.join(HERE, "altitude.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__" and (("altitude" in sys.argv[1:]) or not sys.argv[1:]):
    make_altitude()


def make_raman_pollack():
    """optics.raman_pollack of the reference on its own table (reference/opacities/raman_fortran.txt).
    The fixture carries the table (data), the wavelength grids and the reference's outputs."""
    op = ref_shim.load("optics")
    tab = np.loadtxt(os.path.join(ref_shim.REF_ROOT, "reference", "opacities", "raman_fortran.txt"))
    store = {"table/w": tab[:, 0], "table/f": tab[:, 1]}
    for name, wave in (("vis", np.linspace(0.3, 1.0, 57)), ("wide", 1e4 / np.linspace(500.0, 40000.0, 41))):
        store[name + "/wave"] = wave
        store[name + "/factor"] = op.raman_pollack(4, wave)
    path = os.path.join(HERE, "raman_pollack.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__" and (("raman_pollack" in sys.argv[1:]) or not sys.argv[1:]):
    make_raman_pollack()


def _double_gauss(n_half=4):
    """8-point double-Gauss abscissae/weights on (0,1) in the style of the k-tables: two Gauss-Legendre
    sets on [0, 0.95] and [0.95, 1]."""
    xg, wg = np.polynomial.legendre.leggauss(n_half)
    pts = np.concatenate([0.95 * 0.5 * (xg + 1), 0.95 + 0.05 * 0.5 * (xg + 1)])
    wts = np.concatenate([0.95 * 0.5 * wg, 0.05 * 0.5 * wg])
    return pts, wts


def make_mixing():
    """deq_chem.mix_2_gases / mix_all_gases_gasesfly of the reference on synthetic per-gas ln(kappa)
    tables (monotonic in g, as k-coefficients are), 8 and 4 Gauss points, 2..6 gases."""
    dq = ref_shim.load("deq_chem")
    rng = np.random.default_rng(2024)
    store = {}
    for name, nk, ngas, npres, ntemp, nwno, nlayer in (("g8", 8, 5, 4, 3, 6, 5), ("g4", 4, 3, 3, 3, 5, 4),
                                                        ("two", 8, 2, 3, 2, 4, 3), ("g8many", 8, 9, 3, 3, 3, 3)):
        if nk == 8:
            pts, wts = _double_gauss(4)
        else:
            xg, wg = np.polynomial.legendre.leggauss(nk)
            pts, wts = 0.5 * (xg + 1), 0.5 * wg
        kappas = []
        for g in range(ngas):
            base = -50.0 + 8.0 * rng.random((npres, ntemp, nwno, 1))
            steps = np.cumsum(rng.random((npres, ntemp, nwno, nk)) * rng.choice([0.2, 2.0, 6.0]), axis=3)
            kappas.append(base + steps)
        # one gas with exactly equal coefficients in a bin: ties in the sort
        kappas[-1][0, 0, 0, :] = kappas[-1][0, 0, 0, 0]
        mixes = [10.0 ** (-1.0 - 6.0 * rng.random(nlayer)) for _ in range(ngas)]
        mixes[0] = 0.8 + 0.1 * rng.random(nlayer)
        p_low = rng.integers(0, npres - 1, nlayer)
        t_low = rng.integers(0, ntemp - 1, nlayer)
        p_low[0], t_low[0] = 0, 0
        indices = np.array([p_low, p_low + 1, t_low, t_low + 1])
        out = dq.mix_all_gases_gasesfly(kappas, mixes, pts, wts, indices)
        store[name + "/gauss_pts"], store[name + "/gauss_wts"] = pts, wts
        store[name + "/kappas"] = np.stack(kappas)
        store[name + "/mixes"] = np.stack(mixes)
        store[name + "/indices"] = indices
        store[name + "/kappa_mixed"] = out
        k1, k2 = np.exp(kappas[0][0, 0, 0]), np.exp(kappas[1][0, 0, 0])
        kb, mt = dq.mix_2_gases(k1, k2, mixes[0][0], mixes[1][0], pts, wts)
        store[name + "/pair_kmix"], store[name + "/pair_mix_t"] = kb, np.array(mt)
    path = os.path.join(HERE, "mixing.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__" and (("mixing" in sys.argv[1:]) or not sys.argv[1:]):
    make_mixing()


def _climate_inputs(nlevel=21, nwno=12, ngauss=3, seed=40, cloud_scale=1.0):
    """Per-Gauss-point synthetic planes stacked into the reference's (rows, nwno, ngauss) arrays."""
    nlayer = nlevel - 1
    scs = [syn.make_scene(nlayer, nwno, seed=seed + ig, gas_scale=10.0 ** (ig - 1), cloud_opd=2.0 * cloud_scale)
           for ig in range(ngauss)]
    st = {k: np.ascontiguousarray(np.stack([sc[k] for sc in scs], axis=2))
          for k in PLANES + ("w0_no_raman",)}
    return scs[0], st


def make_climate_fluxes():
    """climate.get_fluxes of the reference (climate.py:1687-1952) with its own namedtuples: the
    correlated-k / level-flux / wavenumber-sum wrapper the climate solver calls every iteration."""
    cl = ref_shim.load("climate")
    store = {}
    for name, nlevel, nwno, ngauss, holes in (("a", 21, 12, 3, False), ("holes", 16, 9, 2, True),
                                              ("g1", 11, 7, 1, False)):
        sc0, st = _climate_inputs(nlevel, nwno, ngauss)
        geo = geometry_1d(5)
        wno = sc0["wno"]
        dwni = np.abs(np.gradient(wno))
        rng = np.random.default_rng(3)
        f0pi = 0.5 + rng.random(nwno)
        atm = cl.Atmosphere_Tuple(None, None, nlevel, sc0["tlevel"], sc0["plevel"], None, None, None, None)
        wed = cl.OpacityWEd_Tuple(st["dtau"], st["tau"], st["w0"], st["cosb"], st["ftau_cld"], st["ftau_ray"],
                                  st["gcos2"], st["w0_no_raman"], None)
        noed = cl.OpacityNoEd_Tuple(st["dtau_og"], st["tau_og"], st["w0_og"], st["cosb_og"])
        sp = cl.ScatteringPhase_Tuple(np.full(nwno, 0.1), 3, 0, 1.0, -1.0, 2.0, -0.5, 1.0)
        dis = cl.Disco_Tuple(5, 1, geo["gweight"], geo["tweight"], geo["ubar0"], geo["ubar1"], 1.0)
        import collections
        Opagrid = collections.namedtuple("Opagrid", ["nwno", "delta_wno", "wno", "ngauss", "gauss_wts"])
        gw = np.array([0.5, 0.3, 0.2][:ngauss])
        gw = gw / gw.sum()
        og = Opagrid(nwno, dwni, wno, ngauss, gw)
        kw = {}
        if holes:
            _, stc = _climate_inputs(nlevel, nwno, ngauss, cloud_scale=0.05)
            kw = dict(do_holes=True, fhole=0.3,
                      hole_OpacityWEd=cl.OpacityWEd_Tuple(stc["dtau"], stc["tau"], stc["w0"], stc["cosb"],
                                                          stc["ftau_cld"], stc["ftau_ray"], stc["gcos2"],
                                                          stc["w0_no_raman"], None),
                      hole_OpacityNoEd=cl.OpacityNoEd_Tuple(stc["dtau_og"], stc["tau_og"], stc["w0_og"],
                                                            stc["cosb_og"]))
            for k, v in stc.items():
                store["%s/clear/%s" % (name, k)] = v
        out = cl.get_fluxes(atm, wed, noed, sp, dis, og, f0pi, True, True, **kw)
        for k, v in st.items():
            store["%s/%s" % (name, k)] = v
        for k, v in dict(tlevel=sc0["tlevel"], plevel=sc0["plevel"], wno=wno, dwni=dwni, f0pi=f0pi, gauss_wts=gw,
                         gweight=geo["gweight"], tweight=geo["tweight"], ubar0=geo["ubar0"],
                         ubar1=geo["ubar1"]).items():
            store["%s/%s" % (name, k)] = v
        for k, v in zip(("flux_net_v_layer", "flux_net_v", "flux_plus_v", "flux_minus_v", "flux_net_ir_layer",
                         "flux_net_ir", "flux_plus_ir", "flux_minus_ir"), out):
            store["%s/out/%s" % (name, k)] = np.asarray(v)
    path = os.path.join(HERE, "climate_fluxes.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__" and (("climate" in sys.argv[1:]) or not sys.argv[1:]):
    make_climate_fluxes()


# ------------------------------------------------------------------------------------------------
# round 5: the remaining option corner of the SH solver and the correlated-k loop around SH / 3-D
# ------------------------------------------------------------------------------------------------
def _sh_reflected(sc, geo, rs, f0, fd, forms, stream, b_top=0.0, flx=0):
    """One get_reflected_SH call of the reference on fresh copies (it writes into f_deltaM)."""
    wsf, wmf, psf, wsr, wmr, psr, sf = forms
    return fl.get_reflected_SH(
        sc["nlevel"], sc["nwno"], geo["numg"], geo["numt"], sc["dtau"].copy(), sc["tau"].copy(), sc["w0"].copy(),
        sc["cosb"].copy(), sc["ftau_cld"].copy(), sc["ftau_ray"].copy(), fd.copy(), sc["dtau_og"].copy(),
        sc["tau_og"].copy(), sc["w0_og"].copy(), sc["cosb_og"].copy(), rs, geo["ubar0"], geo["ubar1"], geo["cos_theta"],
        f0, wsf, wmf, psf, wsr, wmr, psr, TTHG["frac_a"], TTHG["frac_b"], TTHG["frac_c"], TTHG["constant_back"],
        TTHG["constant_forward"], stream, b_top=b_top, flx=flx, single_form=sf)


def make_sh_extra():
    """SH option corner the round-4 fixtures left open: form index 2 ('isotropic', accepted by the reference's approx(),
    justdoit.py:4730-4732; get_reflected_SH falls through with the `ones` weights, fluxes.py:2805-2855) on each of the
    three form arguments, and a non-zero top boundary b_top.  Inputs are those of scene_sh_<name>.npz (not stored again)."""
    sc_all = scenes_1d()
    for name in ("cfg3like", "phase60"):
        sc, geo, rs, f0 = sc_all[name]
        store = {}
        combos = [(2, 2, 2, 1, 1, 1, 0), (2, 0, 0, 1, 1, 1, 0), (0, 2, 0, 1, 1, 1, 0), (0, 0, 2, 1, 1, 1, 0),
                  (2, 1, 2, 0, 0, 0, 1), (1, 2, 0, 1, 0, 1, 1), (2, 2, 2, 0, 1, 0, 0)]
        for stream in (2, 4):
            fd = sc["f_deltaM"].copy()
            if stream == 4:
                fd = sc["cosb_og"] ** 4 if np.any(sc["f_deltaM"]) else fd
            for forms in combos:
                xint, _ = _sh_reflected(sc, geo, rs, f0, fd, forms, stream)
                store["reflsh/s%d_f%d%d%d_r%d%d%d_sf%d/xint" % ((stream,) + forms)] = xint
            for forms, b_top in (((0, 0, 0, 1, 1, 1, 0), 0.3), ((1, 1, 1, 1, 1, 1, 0), 0.05), ((2, 2, 2, 1, 1, 1, 0), 1.0)):
                xint, _ = _sh_reflected(sc, geo, rs, f0, fd, forms, stream, b_top=b_top)
                key = "btop/s%d_f%d%d%d_r%d%d%d_sf%d" % ((stream,) + forms)
                store[key + "/xint"] = xint
                store[key + "/b_top"] = np.array(b_top)
        path = os.path.join(HERE, "sh_extra_%s.npz" % name)
        np.savez_compressed(path, **store)
        print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def make_ck_rt():
    """The correlated-k loop of the reference's picaso() around the SH solvers and around the 3-D solvers
    (justdoit.py:256-307 / 364-380: get_reflected_SH / get_thermal_SH once per Gauss point, `xint_at_top +=
    xint*gauss_wts[ig]`; :488-516: get_reflected_3d / get_thermal_3d per Gauss point on planes with a trailing ngauss
    axis), run here with the reference's own functions.
      sh/...   : the planes the reference's compute_opacity made from ck.npz's 4-Gauss table (ck.npz `de1_s2/*`,
                 `de1_s4/*` -- read from there, not stored again), 5 disk angles at zero phase and 3 x 2 at phase 1.0
      r3d/...  : solver level -- synthetic planes (nlayer|nlevel, nwno, 3, 3, 8) stored whole (small), outputs of the loop
      p3d/...  : pipeline level -- an 8-Gauss premixed table + per-facet temperature / cloud columns through the
                 reference's get_pre_mix_ck + get_continuum + compute_opacity facet by facet (justdoit.py:437-471), then
                 the two loops; inputs and outputs stored (and a few planes of one facet as a spot check)."""
    import types
    import pandas as pd
    optics = ref_shim.load("optics")
    ck = np.load(os.path.join(HERE, "ck.npz"))
    og = np.load(os.path.join(HERE, "optics.npz"))
    wno = og["in/wno"]
    nwno = wno.size
    nlevel = og["in/tlevel"].size
    nlayer = nlevel - 1
    store = {}
    names = ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray", "gcos2", "dtau_og", "tau_og", "w0_og", "cosb_og",
             "w0_no_raman", "f_deltaM")
    gauss_wts = ck["in/gauss_wts"]
    ngauss = gauss_wts.size
    rs = np.linspace(0.05, 0.35, nwno)
    f0 = np.linspace(0.7, 1.4, nwno)
    tlevel = og["in/tlevel"]
    plevel = og["in/plevel_bar"] * 1e6
    store.update({"sh/surf_reflect": rs, "sh/F0PI": f0, "sh/tlevel": tlevel, "sh/plevel": plevel})
    geos = {"g5": geometry_1d(5), "g3x2": geometry_3d(3, 2, 1.0)}
    for gname, geo in geos.items():
        for k, v in geo.items():
            store["sh/%s/geo/%s" % (gname, k)] = np.asarray(v)
    for stream in (2, 4):
        pl = {nm: ck["de1_s%d/%s" % (stream, nm)] for nm in names}
        for gname, geo in geos.items():
            for forms in ((0, 0, 0, 1, 1, 1, 0), (1, 1, 1, 1, 1, 1, 0), (2, 0, 1, 1, 0, 1, 1)):
                xint_at_top = 0
                for ig in range(ngauss):                       # justdoit.py:256-307
                    sc = {nm: np.ascontiguousarray(pl[nm][:, :, ig]) for nm in names}
                    sc.update(nlevel=nlevel, nwno=nwno)
                    xint, _ = _sh_reflected(sc, geo, rs, f0, sc["f_deltaM"], forms, stream)
                    xint_at_top += xint * gauss_wts[ig]
                key = "sh/%s/s%d_f%d%d%d_r%d%d%d_sf%d" % ((gname, stream) + forms)
                store[key + "/xint_at_top"] = xint_at_top
                store[key + "/albedo"] = di.compress_disco(nwno, geo["cos_theta"], xint_at_top, geo["gweight"],
                                                           geo["tweight"], f0)
            for hs in (0, 1):
                flux_at_top = 0
                for ig in range(ngauss):                       # justdoit.py:364-380
                    s_ = {nm: np.ascontiguousarray(pl[nm][:, :, ig]) for nm in names}
                    flux, _ = fl.get_thermal_SH(nlevel, wno, nwno, geo["numg"], geo["numt"], tlevel, s_["dtau"], s_["tau"],
                                                s_["w0"], s_["cosb"], s_["dtau_og"], s_["tau_og"], s_["w0_og"],
                                                s_["w0_no_raman"], s_["cosb_og"], plevel, geo["ubar1"], rs.copy(), stream, hs)
                    flux_at_top += flux * gauss_wts[ig]
                key = "sh/%s/thermal_s%d_hs%d" % (gname, stream, hs)
                store[key + "/flux_at_top"] = flux_at_top
                store[key + "/thermal"] = di.compress_thermal(nwno, flux_at_top, geo["gweight"], geo["tweight"])

    # ---- r3d: the 3-D solvers inside the Gauss loop on stored synthetic planes ----
    ng, nt, ng8 = 3, 3, 8
    xg, wg = np.polynomial.legendre.leggauss(4)
    wts8 = np.concatenate([0.95 * 0.5 * wg, 0.05 * 0.5 * wg])
    geo3 = geometry_3d(ng, nt, np.pi / 3)
    nl3, nw3 = 12, 10
    rng = np.random.default_rng(2025)
    scale = 10.0 ** np.linspace(-1.5, 1.2, ng8)
    planes3 = {k: np.zeros(((nl3 + 1) if k in ("tau", "tau_og") else nl3, nw3, ng, nt, ng8))
               for k in PLANES + ("w0_no_raman",)}
    for g in range(ng):
        for t in range(nt):
            base = syn.make_scene(nl3, nw3, seed=500 + 10 * g + t, cloud_opd=float(rng.uniform(0.05, 3.0)))
            for ig in range(ng8):
                mixed = syn.mix_planes(base["taugas"] * scale[ig], base["tauray"], base["taucld"], base["w0_cld"],
                                       base["g0_cld"])
                for k in planes3:
                    planes3[k][:, :, g, t, ig] = mixed[k]
    p3, tl3 = syn.pressure_temperature(nl3 + 1)
    t3 = np.stack([np.stack([tl3 * (1.0 + 0.1 * rng.uniform(-1, 1)) for _ in range(nt)], axis=1) for _ in range(ng)], axis=1)
    pl3 = np.broadcast_to((p3 * 1e6)[:, None, None], (nl3 + 1, ng, nt)).copy()
    wno3 = syn.wavenumber_grid(nw3)
    f03, rs3 = np.linspace(0.8, 1.3, nw3), np.linspace(0.0, 0.4, nw3)
    for k, v in planes3.items():
        store["r3d/in/" + k] = v
    store.update({"r3d/in/tlevel": t3, "r3d/in/plevel": pl3, "r3d/in/wno": wno3, "r3d/in/F0PI": f03,
                  "r3d/in/surf_reflect": rs3, "r3d/in/gauss_wts": wts8})
    for k, v in geo3.items():
        store["r3d/geo/" + k] = np.asarray(v)

    def loops_3d(pl, wno_, nw_, nlev_, t3_, p3_, rs_, f0_, wts_, key):
        for sp, mp in ((3, 0), (0, 1), (1, 0)):
            xint_at_top = 0
            for ig in range(wts_.size):                       # justdoit.py:488-500
                xint = fl.get_reflected_3d(nlev_, wno_, nw_, ng, nt, *[pl[k][:, :, :, :, ig].copy() for k in PLANES], rs_,
                                           geo3["ubar0"], geo3["ubar1"], geo3["cos_theta"], f0_, sp, mp, TTHG["frac_a"],
                                           TTHG["frac_b"], TTHG["frac_c"], TTHG["constant_back"], TTHG["constant_forward"])
                xint_at_top += xint * wts_[ig]
            store["%s/refl_sp%d_mp%d/xint_at_top" % (key, sp, mp)] = xint_at_top
            store["%s/refl_sp%d_mp%d/albedo" % (key, sp, mp)] = di.compress_disco(
                nw_, geo3["cos_theta"], xint_at_top, geo3["gweight"], geo3["tweight"], f0_)
        for hs in (0, 1):
            flux_at_top = 0
            for ig in range(wts_.size):                       # justdoit.py:502-514
                flux = fl.get_thermal_3d(nlev_, wno_, nw_, ng, nt, t3_, pl["dtau_og"][:, :, :, :, ig].copy(),
                                         pl["w0_no_raman"][:, :, :, :, ig].copy(), pl["cosb_og"][:, :, :, :, ig].copy(),
                                         p3_, geo3["ubar1"], rs_, hs)
                flux_at_top += flux * wts_[ig]
            store["%s/therm_hs%d/flux_at_top" % (key, hs)] = flux_at_top
            store["%s/therm_hs%d/thermal" % (key, hs)] = di.compress_thermal(nw_, flux_at_top, geo3["gweight"],
                                                                             geo3["tweight"])
    loops_3d(planes3, wno3, nw3, nl3 + 1, t3, pl3, rs3, f03, wts8, "r3d")

    # ---- p3d: table -> per-facet compute_opacity -> the two loops ----
    temps = np.array([100.0, 250.0, 600.0, 1200.0, 2600.0])
    press = np.array([1e-6, 1e-4, 1e-2, 1.0, 30.0, 300.0])
    nc_p = np.array([6, 6, 6, 5, 4])
    pressures = np.concatenate([press[:n] for n in nc_p])
    temps_flat = np.concatenate([[t] * n for t, n in zip(temps, nc_p)])
    rng = np.random.default_rng(888)
    kappa = np.zeros((press.size, temps.size, nwno, ng8))
    for ip, p_ in enumerate(press):
        for it, t_ in enumerate(temps):
            base = (-26.0 + 2.0 * np.sin(wno / 2500.0) + 0.5 * np.log10(p_) + 0.9 * np.log10(t_ / 300.0))
            steps = np.cumsum(0.35 + 0.3 * rng.random((nwno, ng8)), axis=1)
            kappa[ip, it] = np.log(10.0) * (base[:, None] - 1.0 + steps)
    cia_temps = np.array([75.0, 200.0, 500.0, 1000.0, 2000.0, 4000.0])
    mixkeys = ("H2", "He", "H2O", "CH4")
    mix = {k: og["in/mix/" + k] for k in mixkeys}
    gravity = float(og["in/gravity"])
    weights = _ref_weights(mixkeys)
    plevel_bar = og["in/plevel_bar"]
    tfac = 1.0 + 0.12 * np.sin(1.0 + np.arange(ng * nt)).reshape(ng, nt)             # per-facet temperature scale
    cfac = (0.2 + 1.6 * rng.random((ng, nt)))                                        # per-facet cloud optical-depth scale
    t3p = og["in/tlevel"][:, None, None] * tfac[None]
    db = os.path.join(HERE, "synthetic_opacities.db")

    def make_atm(g, t):
        atm = types.SimpleNamespace()
        atm.c = types.SimpleNamespace(nlayer=nlayer, nlevel=nlevel, pconv=1e6, k_b=1.380649e-16,
                                      amu=1.66053906660e-24, rgas=8.31446261815324)
        atm.planet = types.SimpleNamespace(gravity=gravity)
        p = plevel_bar * 1e6
        tl = t3p[:, g, t]
        atm.level = {"pressure": p, "temperature": tl}
        lay_mix = pd.DataFrame({k: 0.5 * (v[1:] + v[:-1]) for k, v in mix.items()})
        mmw_lvl = sum(mix[k] * weights[k] for k in mix)
        atm.layer = {"pressure": np.sqrt(p[1:] * p[:-1]), "temperature": 0.5 * (tl[1:] + tl[:-1]),
                     "mmw": 0.5 * (mmw_lvl[1:] + mmw_lvl[:-1]), "colden": _ref_colden(p, tl, mmw_lvl, gravity),
                     "electrons": np.zeros(nlayer), "mixingratios": lay_mix,
                     "cloud": {"opd": og["in/cld_opd"] * cfac[g, t], "w0": og["in/cld_w0"].copy(),
                               "g0": og["in/cld_g0"].copy()}}
        atm.molecules = np.array(["H2O", "CH4", "H2"])
        atm.continuum_molecules = [["H2", "H2"], ["H2", "He"], ["H2", "CH4"]]
        atm.rayleigh_molecules = ["H2", "He", "CH4", "H2O"]
        return atm

    opa = object.__new__(optics.RetrieveCKs)
    opa.pressures, opa.temps, opa.nc_p, opa.kappa = pressures, temps_flat, nc_p, kappa
    opa.continuum_db, opa.cia_temps = db, cia_temps
    opa.wno, opa.nwno, opa.ngauss, opa.gauss_wts = wno, nwno, ng8, wts8
    rayleigh = ref_shim.load("rayleigh")
    ray = rayleigh.Rayleigh(wno)
    opa.rayleigh_opa = {m: ray.compute_sigma(m) for m in ("H2", "He", "CH4", "H2O")}
    pl3p = {k: np.zeros(((nlevel if k in ("tau", "tau_og") else nlayer), nwno, ng, nt, ng8)) for k in names}
    for g in range(ng):
        for t in range(nt):                                    # justdoit.py:437-471
            atm = make_atm(g, t)
            opa.get_pre_mix_ck(atm)
            opa.get_continuum(atm)
            out = optics.compute_opacity(atm, opa, ngauss=ng8, stream=2, delta_eddington=True, raman=2, test_mode=None)
            for nm, arr in zip(names, out):
                pl3p[nm][:, :, g, t, :] = arr
    p3p = np.broadcast_to((plevel_bar * 1e6)[:, None, None], (nlevel, ng, nt)).copy()
    store.update({"p3d/in/press": press, "p3d/in/temps": temps, "p3d/in/nc_p": nc_p, "p3d/in/kappa": kappa,
                  "p3d/in/gauss_wts": wts8, "p3d/in/cia_temps": cia_temps, "p3d/in/tlevel": t3p,
                  "p3d/in/cloud_scale": cfac, "p3d/in/surf_reflect": rs, "p3d/in/F0PI": f0})
    for nm in ("dtau", "w0", "tau_og", "w0_no_raman", "cosb_og"):
        store["p3d/facet_2_1/" + nm] = pl3p[nm][:, :, 2, 1, :]
    loops_3d(pl3p, wno, nwno, nlevel, t3p, p3p, rs, f0, wts8, "p3d")
    path = os.path.join(HERE, "ck_rt.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__" and (("sh_extra" in sys.argv[1:]) or not sys.argv[1:]):
    make_sh_extra()
if __name__ == "__main__" and (("ck_rt" in sys.argv[1:]) or not sys.argv[1:]):
    make_ck_rt()


def make_planck():
    """fluxes.blackbody / blackbody_integrated / chapman / tidal_flux (fluxes.py:1609-1680, 3671-3751) on seeded
    inputs that include an overflowing exponential (cold level, short wavelength)."""
    import collections
    rng = np.random.default_rng(77)
    t = np.concatenate([[40.0, 68.0], np.sort(rng.uniform(75.0, 3200.0, 19))])
    wno = np.sort(rng.uniform(30.0, 33000.0, 57))[::-1].copy()
    dwno = np.abs(np.gradient(wno)) * rng.uniform(0.6, 1.4, wno.size)
    store = dict(t=t, wno=wno, dwno=dwno, w_cm=1.0 / wno)
    with np.errstate(over="ignore"):
        store["blackbody"] = fl.blackbody(t, 1.0 / wno)
        store["blackbody_integrated"] = fl.blackbody_integrated(t, wno, dwno)
    nlevel = 31
    pressure = np.logspace(-6, 2, nlevel)
    col_den = rng.uniform(0.5, 2.0, nlevel - 1) * np.diff(pressure) * 1e3
    store.update(pressure=pressure, col_den=col_den)
    Bundle = collections.namedtuple("InjectionBundle", "inject_beam beam_profile pm hratio wave_in")
    beam = rng.uniform(0.0, 5.0e3, nlevel)
    store["beam_profile"] = beam
    store["chapman"] = np.array([fl.chapman(p, 0.01, 1.7) for p in pressure])
    store["tidal_chapman"] = fl.tidal_flux(450.0, nlevel, pressure, col_den, Bundle(False, None, 0.01, 1.7, 2.5e6))
    store["tidal_beam"] = fl.tidal_flux(450.0, nlevel, pressure, col_den, Bundle(True, beam, 0.0, 0.0, 0.0))
    path = os.path.join(HERE, "planck.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__" and (("planck" in sys.argv[1:]) or not sys.argv[1:]):
    make_planck()
