"""Second matrix: k-tables x dimension x solver x legs x cloud form x delta_eddington x raman: nothing may raise or go non-finite."""
import itertools, os, sys, tempfile, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
warnings.simplefilter("ignore")
from picaso_amd import _lib
from picaso_amd import justdoit as jdi
from picaso_amd import optics as px
ctx = _lib.context(0)
nlevel = 31
temps, press = [100.0, 300.0, 700.0, 1500.0, 3000.0], [1e-6, 1e-4, 1e-2, 1e-1, 1.0, 10.0, 100.0, 500.0]
cia_t = [75.0, 500.0, 4000.0]
def mono(nwno=300):
    wno = np.linspace(2000.0, 33333.0, nwno)
    pt = [(i + 1, p, t) for i, (t, p) in enumerate((t, p) for t in temps for p in press)]
    molecular = {m: {i: 10.0 ** (-24.0 + 2.0 * np.sin(wno / 2500.0 + k) + 0.4 * np.log10(p)) for (i, p, t) in pt} for k, m in enumerate(("H2O", "CH4"))}
    continuum = {pr: {t: 10.0 ** (-7.0 + np.cos(wno / 4000.0 + k)) for t in cia_t} for k, pr in enumerate(("H2H2", "H2He"))}
    ray = {m: 1e-27 * (wno / 1e4) ** 4 for m in ("H2", "He")}
    return px.RetrieveOpacities(wno, pt, molecular, continuum, cia_t, rayleigh_opa=ray, query_method="linear", ctx=ctx)
def ck(nb=120, nk=8):
    wck = np.linspace(40.0, 28000.0, nb)
    xg, wg = np.polynomial.legendre.leggauss(4)
    gpts = np.concatenate([0.95 * 0.5 * (xg + 1), 0.95 + 0.05 * 0.5 * (xg + 1)]); gwts = np.concatenate([0.95 * 0.5 * wg, 0.05 * 0.5 * wg])
    tk, pk = np.array(temps), np.array(press)
    lnk = np.log(10.0) * (-26.0 + 2.0 * np.sin(wck / 2500.0)[None, None, :, None] + 0.5 * np.log10(pk)[:, None, None, None]
                          + 0.9 * np.log10(tk / 300.0)[None, :, None, None] + 0.6 * np.arange(nk)[None, None, None, :])
    cont = {pr: {t: 10.0 ** (-7.0 + np.cos(wck / 4000.0 + k) + 0.3 * np.log10(t / 300.0)) for t in cia_t} for k, pr in enumerate(("H2H2", "H2He"))}
    return px.RetrieveCKs(wck, gwts, np.tile(pk, tk.size), np.repeat(tk, pk.size), np.full(tk.size, pk.size), lnk, continuum=cont, cia_temps=cia_t,
                          rayleigh_opa={m: 1e-27 * (wck / 1e4) ** 4 for m in ("H2", "He")}, gauss_pts=gpts, ctx=ctx)
def ck_fly(nb=120, nk=8):
    """per-gas k-tables mixed on the fly (resort-rebin, deq_chem.py:334)"""
    wck = np.linspace(40.0, 28000.0, nb)
    xg, wg = np.polynomial.legendre.leggauss(4)
    gpts = np.concatenate([0.95 * 0.5 * (xg + 1), 0.95 + 0.05 * 0.5 * (xg + 1)]); gwts = np.concatenate([0.95 * 0.5 * wg, 0.05 * 0.5 * wg])
    tk, pk = np.array(temps), np.array(press)
    def tab(j):
        return np.log(10.0) * (-26.0 + 2.0 * np.sin(wck / 2500.0 + j)[None, None, :, None] + 0.5 * np.log10(pk)[:, None, None, None]
                               + 0.9 * np.log10(tk / 300.0)[None, :, None, None] + 0.6 * np.arange(nk)[None, None, None, :])
    cont = {pr: {t: 10.0 ** (-7.0 + np.cos(wck / 4000.0 + k) + 0.3 * np.log10(t / 300.0)) for t in cia_t} for k, pr in enumerate(("H2H2", "H2He"))}
    return px.RetrieveCKs(wck, gwts, np.tile(pk, tk.size), np.repeat(tk, pk.size), np.full(tk.size, pk.size), None, continuum=cont, cia_temps=cia_t,
                          rayleigh_opa={m: 1e-27 * (wck / 1e4) ** 4 for m in ("H2", "He")}, kappas={"H2O": tab(0), "CH4": tab(1), "H2": tab(2) - 8.0},
                          gauss_pts=gpts, on_fly=True, ctx=ctx)
OPAS = {"mono": mono(), "ck": ck(), "ck_fly": ck_fly()}
d = tempfile.mkdtemp(); os.makedirs(os.path.join(d, "opacities"))
wn = np.round(np.linspace(30.0, 34000.0, 196)[::-1], 2)
with open(os.path.join(d, "opacities", "wave_EGP.dat"), "w") as fh:
    fh.write("   i   micron.    wavenumber idum     idum1    idum2     idum3\n")
    for i, w in enumerate(wn):
        fh.write("%4d %9.3f %9.2f %8.2f- %7.2f %9.3f %9.3f\n" % (i + 1, 1e4 / w, w, w - 1, w + 1, 2.0, w))
os.environ["picaso_refdata"] = d
plev = np.logspace(-6, 2, nlevel)
prof = {"pressure": plev, "temperature": 150.0 + 1200.0 * ((np.log10(plev) + 6) / 8) ** 2, "H2": np.full(nlevel, 0.84),
        "He": np.full(nlevel, 0.155), "H2O": np.full(nlevel, 1e-3), "CH4": np.full(nlevel, 5e-4)}
pert = 1.0 + 0.1 * np.cos(np.arange(16).reshape(4, 4))
CALCS = ["reflected", "thermal", "reflected+thermal", "reflected+thermal+transmission", "transmission"]
nbad = ntot = 0
for oname, dim, sh, calc, cloud, de, full in itertools.product(("mono", "ck", "ck_fly"), ("1d", "3d"), (False, True), CALCS, ("none", "own-grid"), (True, False), (False, True)):
    opa = OPAS[oname]; wno = opa.wno; nwno = wno.size
    if dim == "3d" and "transmission" in calc:
        continue                                   # the reference has no 3-D transmission branch: a clean error here (checked below once)
    c = jdi.inputs()
    if dim == "3d":
        c.phase_angle(np.pi / 3, num_gangle=4, num_tangle=4)
        c.atmosphere_3d(dict(prof, temperature=prof["temperature"][:, None, None] * pert[None]))
    else:
        c.phase_angle(0)
        c.atmosphere(df=prof)
    c.star(relative_flux=1.0 + 0.2 * np.cos(wno / 900.0), radius=6.9e10, semi_major=7.5e12)
    c.gravity(radius=7.1e9, mass=1.9e30)
    if cloud == "own-grid":
        box = np.zeros((nlevel - 1, 196)); box[15:20] = 0.3
        df = {"opd": box, "w0": np.where(box > 0, 0.95, 0.0), "g0": np.where(box > 0, 0.6, 0.0), "wavenumber": np.linspace(wno[0], wno[-1], 196)}
        if dim == "3d":
            c.clouds_3d(df=df)
        else:
            c.clouds(g0=[0.8], w0=[0.95], opd=[1.5], p=[0.0], dp=[1.5])         # a box cloud on the 196-point grid file
    tag = "%s | %s | %s | %s | cloud %s | delta_eddington %d | full %d" % (oname, dim, "SH4" if sh else "toon", calc, cloud, de, full)
    ntot += 1
    try:
        c.approx(raman="none", delta_eddington=de, **({"rt_method": "SH", "stream": 4} if sh else {}))
        r = c.spectrum(opa, calculation=calc, dimension=dim, full_output=full)
        for k in ("albedo", "thermal", "transit_depth"):
            if k in r and not np.all(np.isfinite(r[k])):
                print("NONFINITE %s in %s" % (k, tag)); nbad += 1
    except Exception as e:
        print("RAISED   %s: %s: %s" % (tag, type(e).__name__, str(e)[:110])); nbad += 1
print("combinations %d, problems %d" % (ntot, nbad))
