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
