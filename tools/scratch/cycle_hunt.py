"""Which objects keep DeviceArrays alive until the cyclic collector runs?  (run on the GPU box)"""
import gc, os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from picaso_amd import _lib, device
from picaso_amd import justdoit as jdi
from picaso_amd import optics as px
ctx = _lib.context(0)
nwno, nlevel = 5000, 61
wno = np.linspace(2000.0, 33333.0, nwno)
temps, press = [100.0, 300.0, 700.0, 1500.0, 3000.0], [1e-6, 1e-4, 1e-2, 1e-1, 1.0, 10.0, 100.0, 500.0]
pt = [(i + 1, p, t) for i, (t, p) in enumerate((t, p) for t in temps for p in press)]
mols = ["H2O", "CH4"]
molecular = {m: {i: 10.0 ** (-24.0 + 2.0 * np.sin(wno / 2500.0 + k) + 0.4 * np.log10(p)) for (i, p, t) in pt} for k, m in enumerate(mols)}
cia_t = [75.0, 500.0, 4000.0]
continuum = {pr: {t: 10.0 ** (-7.0 + np.cos(wno / 4000.0 + k)) for t in cia_t} for k, pr in enumerate(("H2H2", "H2He"))}
ray = {m: 1e-27 * (wno / 1e4) ** 4 for m in ("H2", "He")}
opa = px.RetrieveOpacities(wno, pt, molecular, continuum, cia_t, rayleigh_opa=ray, query_method="linear", ctx=ctx)
plev = np.logspace(-6, 2, nlevel)
rng = np.random.default_rng(1)

def run(kind):
    prof = {"pressure": plev, "temperature": (150.0 + 1200.0 * ((np.log10(plev) + 6) / 8) ** 2) * (1 + 0.1 * rng.random()),
            "H2": np.full(nlevel, 0.84), "He": np.full(nlevel, 0.155), "H2O": np.full(nlevel, 1e-3), "CH4": np.full(nlevel, 5e-4)}
    case = jdi.inputs()
    if kind == 4: case.phase_angle(0.7, num_gangle=6, num_tangle=6)
    else: case.phase_angle(0)
    case.gravity(gravity=2500.0)
    case.atmosphere(df=prof)
    akw = {"raman": "none"}
    if kind == 3: akw.update(rt_method="SH", stream=4)
    if kind == 5: akw.update(get_lvl_flux=True)
    case.approx(**akw)
    calc = "reflected+thermal"
    if kind in (1, 2, 3):
        shp = (nlevel - 1, nwno); opd = np.zeros(shp); opd[30:36] = 0.3
        case.clouds(df={"opd": opd, "w0": np.full(shp, 0.9), "g0": np.full(shp, 0.5)})
    if kind == 2:
        case.star(relative_flux=1.0 + 0.2 * np.cos(wno / 900.0), radius=6.9e10, semi_major=7.5e12)
        case.gravity(radius=7.1e9, mass=1.9e30)
        calc = "reflected+thermal+transmission"
    return case.spectrum(opa, calculation=calc)

def ndev():
    return sum(1 for o in gc.get_objects() if isinstance(o, device.DeviceArray))

for kind in range(6):
    run(kind); gc.collect()
    gc.disable()
    base = ndev()
    r = run(kind); del r
    after = ndev()
    # who holds the extra ones?
    extra = [o for o in gc.get_objects() if isinstance(o, device.DeviceArray)]
    holders = collections.Counter()
    for o in extra[-(after - base):] if after > base else []:
        for ref in gc.get_referrers(o):
            if ref is extra or isinstance(ref, type(sys._getframe())): continue
            holders[type(ref).__name__ + (":" + ",".join(sorted(k for k in ref.keys() if isinstance(k, str))[:8]) if isinstance(ref, dict) else "")] += 1
    n = gc.collect()
    print("kind", kind, "device arrays before", base, "after one call", after, "after gc", ndev(), "collected", n)
    for h, c in holders.most_common(6): print("     held by", c, h[:200])
    gc.enable()
