"""phase_curve: reflected / thermal x monochromatic / k-tables x cloud map none / one / per phase x devices None / [0, 0]:
nothing raises, everything finite, the devices form equals the plain one bit for bit."""
import itertools, os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools", "scratch"))
warnings.simplefilter("ignore")
src = open(os.path.join(ROOT, "tools", "scratch", "matrix_probe2.py")).read().split("plev = np.logspace")[0]
exec(src)
from picaso_amd import justdoit as jdi
plev = np.logspace(-6, 2, nlevel)
prof = {"pressure": plev, "temperature": 150.0 + 1200.0 * ((np.log10(plev) + 6) / 8) ** 2, "H2": np.full(nlevel, 0.84),
        "He": np.full(nlevel, 0.155), "H2O": np.full(nlevel, 1e-3), "CH4": np.full(nlevel, 5e-4)}
pert = 1.0 + 0.1 * np.cos(np.arange(16).reshape(4, 4))
phases = [0.4, 2.0, 4.1]
bad = tot = 0
for oname, kind, cloud in itertools.product(("mono", "ck"), ("reflected", "thermal"), ("none", "one", "per-phase")):
    opa = OPAS[oname]; wno = opa.wno
    def build():
        pc = jdi.inputs()
        pc.phase_curve_geometry(kind, phases, num_gangle=4, num_tangle=4)
        pc.gravity(gravity=2500.0)
        pc.star(relative_flux=1.0 + 0.2 * np.cos(wno / 900.0), radius=6.9e10, semi_major=7.5e12)
        pc.atmosphere_4d([dict(prof, temperature=prof["temperature"][:, None, None] * (pert[None] + 0.01 * k)) for k in range(len(phases))])
        pc.approx(raman="none")
        kw = {}
        box = np.zeros((nlevel - 1, 196)); box[15:20] = 0.3
        def tab(scale):
            b = box * scale
            return {"opd": b, "w0": np.where(b > 0, 0.95, 0.0), "g0": np.where(b > 0, 0.6, 0.0), "wavenumber": np.linspace(wno[0], wno[-1], 196)}
        if cloud == "one":
            pc.clouds_3d(df=tab(1.0))
        elif cloud == "per-phase":
            kw["clouds_by_phase"] = [tab(1.0 + 0.2 * k) for k in range(len(phases))]
        return pc, kw
    tag = "%s | %s | cloud %s" % (oname, kind, cloud)
    tot += 1
    try:
        pc, kw = build()
        a = pc.phase_curve(opa, **kw)
        key = "albedo" if kind == "reflected" else "thermal"
        if not all(np.all(np.isfinite(v[key])) for v in a.values()):
            print("NONFINITE", tag); bad += 1
        pc2, kw2 = build()
        b = pc2.phase_curve(opa, devices=[0, 0], **kw2)
        if not all(np.array_equal(a[p][key], b[p][key]) for p in a):
            print("DEVICES DIFFER", tag); bad += 1
    except Exception as e:
        import traceback
        print("RAISED  %s: %s: %s" % (tag, type(e).__name__, str(e)[:160])); bad += 1
print("combinations %d, problems %d" % (tot, bad))
