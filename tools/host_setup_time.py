#!/usr/bin/env python
"""Host-side set-up of one spectrum() call, timed stage by stage WITHOUT a GPU: ATMSETUP, table-row search
(get_opacities), per-layer coefficients, job struct.  The device tables are stand-ins (the stages never touch them);
the arithmetic is what the product runs.  Also prints a checksum of every stage's result, so that a change meant to
save time can be seen not to change a bit.      python tools/host_setup_time.py [ncalls]"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from picaso_amd import optics as px, justdoit as jdi, driver as drv


class _Dev:                      # DeviceArray stand-in
    addr = 0

    def __init__(self, *a, **k):
        pass

    @classmethod
    def from_host(cls, a, ctx=None):
        return cls()


px.DeviceArray = _Dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nwno, nlevel = 64, 91
wno = np.linspace(2000.0, 33333.0, nwno)
temps, press = [100.0, 300.0, 700.0, 1500.0, 3000.0], [1e-6, 1e-4, 1e-2, 1e-1, 1.0, 10.0, 100.0, 500.0]
pt = [(i + 1, p, t) for i, (t, p) in enumerate((t, p) for t in temps for p in press)]
mols = ["H2O", "CH4", "CO", "NH3", "H2"]
molecular = {m: {i: np.ones(nwno) for (i, p, t) in pt} for m in mols}
cia_t = [75.0, 200.0, 500.0, 1000.0, 2000.0, 4000.0]
continuum = {pr: {t: np.ones(nwno) for t in cia_t} for pr in ("H2H2", "H2He")}
ray = {m: np.ones(nwno) for m in ("H2", "He")}
opa = px.RetrieveOpacities(wno, pt, molecular, continuum, cia_t, rayleigh_opa=ray, query_method="linear", ctx=object())
plev = np.logspace(-6, 2, nlevel)
rng = np.random.default_rng(1)


def profile(k):
    return {"pressure": plev, "temperature": (150.0 + 1200.0 * ((np.log10(plev) + 6) / 8) ** 2) * (1.0 + 0.01 * k),
            "H2": np.full(nlevel, 0.84), "He": np.full(nlevel, 0.155), "H2O": np.full(nlevel, 1e-3) * (1 + 0.1 * k),
            "CH4": np.full(nlevel, 5e-4), "CO": np.full(nlevel, 1e-4), "NH3": np.full(nlevel, 1e-5)}


case = jdi.inputs()
case.phase_angle(0)
case.gravity(gravity=2500.0)
case.approx(raman="none")
h = hashlib.sha256()
acc = {"atm": 0.0, "get_opacities": 0.0, "factors": 0.0, "job": 0.0}
for k in range(n):
    case.atmosphere(df=profile(k % 7))
    inp = case.inputs
    t0 = time.perf_counter()
    atm = jdi._setup_atmosphere(inp, opa, wno)
    t1 = time.perf_counter()
    opa.get_opacities(atm, exclude_mol=inp["atmosphere"]["exclude_mol"])
    t2 = time.perf_counter()
    fac = px._layer_factors(atm, opa)
    t3 = time.perf_counter()
    geom = inp["disco"]
    common = inp["approx"]["rt_params"]["common"]
    toon = inp["approx"]["rt_params"]["toon"]
    job, keep = drv.make_job(atm.c.nlayer, opa._plan, fac, True, atm.c.nlayer, common["stream"], common["delta_eddington"], True,
                             True, geom["num_gangle"], geom["num_tangle"], geom["ubar0"], geom["ubar1"], geom["cos_theta"],
                             geom["gweight"], geom["tweight"], toon["single_phase"], toon["multi_phase"],
                             toon["toon_coefficients"], *common["TTHG_params"]["fraction"],
                             common["TTHG_params"]["constant_back"], common["TTHG_params"]["constant_forward"], 0.0,
                             atm.level["temperature"], atm.level["pressure"], atm.hard_surface)
    t4 = time.perf_counter()
    acc["atm"] += t1 - t0; acc["get_opacities"] += t2 - t1; acc["factors"] += t3 - t2; acc["job"] += t4 - t3
    if k < 7:
        for a in (atm.level["z"], atm.level["dz"], atm.level["scale_height"], atm.level["den"], atm.level["mmw"],
                  atm.layer["colden"], atm.layer["gravity"], atm.layer["mmw"], atm.layer["pressure"], atm.layer["temperature"],
                  opa._plan["rows"], opa._plan["wts"], opa._plan["cia_rows"], opa._plan["fac"], fac[0], fac[1], fac[3],
                  keep["rows"], keep["wts"], keep["mol_fac"], keep["cont_rows"], keep["cont_fac"], keep["ray_fac"]):
            h.update(np.ascontiguousarray(a).tobytes())
        h.update(repr((list(atm.molecules), atm.continuum_molecules, atm.rayleigh_molecules, fac[2],
                       sorted(atm.layer["mixingratios"]))).encode())
print("  ".join("%s %.1f us" % (k, 1e6 * v / n) for k, v in acc.items()), " total %.1f us" % (1e6 * sum(acc.values()) / n))
print("checksum", h.hexdigest()[:16])
