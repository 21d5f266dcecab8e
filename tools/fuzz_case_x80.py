#!/usr/bin/env python
"""A fuzz case that exceeded its tolerance (tests/test_fuzz_gpu.py, PICASO_FUZZ_DUMP=dir writes inputs and both
results): where do the HIP result and the C oracle sit relative to the reference evaluated in fp64 and in x87
extended precision?  Run in the build container (imports the reference under tools/ref_shim.py).

    python tools/fuzz_case_x80.py gpurun_out/fuzz/dump/*.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_shim  # noqa: E402
from helpers import PLANES, lvl_err  # noqa: E402

fl = ref_shim.load("fluxes")
TTHG = (1.0, -1.0, 2.0, -0.5, 1.0)


def extended(fn, *args, **kw):
    L = np.longdouble
    wide = [a.astype(L) if isinstance(a, np.ndarray) and a.dtype == np.float64 else a for a in args]
    orig = fl.zeros
    fl.zeros = lambda *a, **k: np.zeros(*a, dtype=L, **k)
    try:
        return fn(*wide, **kw)
    finally:
        fl.zeros = orig


for path in sys.argv[1:]:
    d = np.load(path)
    name = os.path.basename(path)
    if name.startswith("reflected_lvl"):
        nlayer, nwno = d["dtau"].shape
        ng, nt = d["u0"].shape
        sp, mp, tc = (int(x) for x in d["opts"])
        rs = np.zeros(nwno) + d["rs"]
        args = (nlayer + 1, d["wno"], nwno, ng, nt, *[d[k].copy() for k in PLANES], rs, d["u0"], d["u1"], float(d["ct"]),
                d["f0"], sp, mp, *TTHG)
        kw = dict(get_toa_intensity=1, get_lvl_flux=1, toon_coefficients=tc, b_top=float(d["b_top"]))
        _, l64 = fl.get_reflected_1d(*args, **kw)
        _, l80 = extended(fl.get_reflected_1d, *args, **kw)
        l80 = [np.asarray(a, dtype=np.float64) for a in l80]
        lg, lo = list(d["lg"]), list(d["lo"])
        print("%s  layers %d  max dtau %.1f" % (name, nlayer, float(d["dtau_og"].max())))
        print("   C oracle vs reference fp64  %.2e" % lvl_err(lo, l64))
        print("   reference fp64 vs its x80   %.2e   <- the reference's own rounding on this scene" % lvl_err(l64, l80))
        print("   HIP vs reference fp64       %.2e" % lvl_err(lg, l64))
        print("   HIP vs reference x80        %.2e" % lvl_err(lg, l80))
    else:
        nlayer, nwno = d["dtau_og"].shape
        ng, nt = d["u1"].shape
        rs = np.zeros(nwno) + d["rs"]
        args = (nlayer + 1, d["wno"], nwno, ng, nt, d["tlevel"], d["dtau_og"].copy(), d["w0_no_raman"].copy(),
                d["cosb_og"].copy(), d["plevel"], d["u1"], rs, int(d["hard"]), d["dw"], int(d["calc"]))
        f64, _ = fl.get_thermal_1d(*args)
        f80, _ = extended(fl.get_thermal_1d, *args)
        f80 = np.asarray(f80, dtype=np.float64)
        sc = 1e-4 * np.abs(f64).max()

        def err(a, b):
            return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), sc)))
        print("%s  layers %d  max dtau %.1f" % (name, nlayer, float(d["dtau_og"].max())))
        print("   C oracle vs reference fp64  %.2e" % err(d["fo"], f64))
        print("   reference fp64 vs its x80   %.2e   <- the reference's own rounding on this scene" % err(f64, f80))
        print("   HIP vs reference fp64       %.2e" % err(d["fg"], f64))
        print("   HIP vs reference x80        %.2e" % err(d["fg"], f80))
