#!/usr/bin/env python
"""Host-side profile of climate.get_fluxes at the climate tables' shape (91 levels, 661 bins x 8 Gauss points,
resident planes) -- run on the GPU box."""
import cProfile, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from picaso_amd import _lib, climate as pc, disco, resident, synthetic as syn
from picaso_amd.device import DeviceArray

ctx = _lib.context(0)
nlev, nw, ngq = 91, 661, 8
scs = [syn.make_scene(nlev - 1, nw, seed=70 + ig, gas_scale=10.0 ** (0.5 * ig - 2)) for ig in range(ngq)]
keys = resident.REFLECTED_PLANES + ("w0_no_raman",)
st = {k: np.ascontiguousarray(np.stack([sc[k] for sc in scs], axis=2)) for k in keys}
xg, wg = np.polynomial.legendre.leggauss(ngq)
g, gw, t, tw = disco.get_angles_1d(5)
u0, u1, ct, _, _ = disco.compute_disco(5, 1, g, t, 0.0)
wno_c = scs[0]["wno"]
atm_t = pc.Atmosphere_Tuple(None, None, nlev, scs[0]["tlevel"], scs[0]["plevel"], None, None, None, None)
sp_t = pc.ScatteringPhase_Tuple(np.zeros(nw), 3, 0, 1.0, -1.0, 2.0, -0.5, 1.0)
dis_t = pc.Disco_Tuple(5, 1, gw, tw, u0, u1, 1.0)
og_t = pc.Opagrid_Tuple(nw, np.abs(np.gradient(wno_c)), wno_c, ngq, 0.5 * wg)
conv = lambda a: DeviceArray.from_host(a, ctx)
wd = pc.OpacityWEd_Tuple(*[conv(st[k]) for k in ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray", "gcos2", "w0_no_raman")], None)
nd = pc.OpacityNoEd_Tuple(*[conv(st[k]) for k in ("dtau_og", "tau_og", "w0_og", "cosb_og")])
for _ in range(5):
    pc.get_fluxes(atm_t, wd, nd, sp_t, dis_t, og_t, np.ones(nw), True, True)
ts = []
for _ in range(30):
    t0 = time.perf_counter()
    pc.get_fluxes(atm_t, wd, nd, sp_t, dis_t, og_t, np.ones(nw), True, True)
    ts.append(time.perf_counter() - t0)
print("get_fluxes min %.3f ms median %.3f ms" % (1e3 * min(ts), 1e3 * float(np.median(ts))))
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    pc.get_fluxes(atm_t, wd, nd, sp_t, dis_t, og_t, np.ones(nw), True, True)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
