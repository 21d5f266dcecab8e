"""Bad arguments straight at the C ABI: every call must come back with rc != 0 and a message, never crash the process.
Each case runs in a child process so that a crash is seen as an exit status."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CASES = ["refl_nlevel1", "refl_nwno0", "refl_numg0", "refl_null_dtau", "refl_null_out", "refl_bad_single_phase", "refl_numg_huge",
         "therm_nlevel1", "therm_nwno_neg", "therm_null_tlevel", "therm_calc_type9", "transit_nlevel1", "transit_null",
         "bb_ntemp0", "bb_null", "disco_ng0", "disco_null", "sh_stream3", "sh_null"]

def child(name):
    from picaso_amd import _lib
    lib, ctx = _lib.load(), _lib.context(0)
    P = ctypes.POINTER(ctypes.c_double)
    nl, nw, ng = 5, 64, 5
    def arr(*shape, v=0.1):
        return np.full(shape, v)
    def p(a):
        return None if a is None else a.ctypes.data_as(P)
    ci, cd = ctypes.c_int, ctypes.c_double
    lay, lev, vec = arr(nl - 1, nw), arr(nl, nw), arr(nw)
    u = arr(ng, 1, v=0.5)
    x = arr(ng, 1, nw)
    def refl(nlevel=nl, nwno=nw, numg=ng, dtau=lay, out=x, sp=3):
        return lib.picaso_get_reflected_1d(ctx, ci(nlevel), p(vec), ci(nwno), ci(numg), ci(1), p(dtau), p(lev), p(lay), p(lay), p(lay),
            p(lay), p(lay), p(lay), p(lev), p(lay), p(lay), p(vec), p(u), p(u), cd(1.0), p(vec), ci(sp), ci(0), cd(1.0), cd(-1.0),
            cd(2.0), cd(-0.5), cd(1.0), ci(1), ci(0), ci(0), cd(0.0), p(out), None, None, None, None)
    def therm(nlevel=nl, nwno=nw, tl=arr(nl, v=500.0), ct=0):
        return lib.picaso_get_thermal_1d(ctx, ci(nlevel), p(arr(nw, v=1000.0)), ci(nwno), ci(ng), ci(1), p(tl), p(lay), p(lay), p(lay),
            p(arr(nl, v=1.0) * np.arange(1, nl + 1)), p(u), p(vec), ci(0), p(vec), ci(ct), p(x), None, None, None, None)
    def transit(nlevel=nl, z=arr(nl, v=1e9)):
        return lib.picaso_get_transit_1d(ctx, p(z), p(arr(nl, v=1e5)), ci(nlevel), ci(nw), cd(7e10), p(arr(nl - 1, v=2.3)), cd(1.38e-16),
            cd(1.66e-24), p(arr(nl - 1, v=1e5)), p(arr(nl - 1, v=500.0)), p(arr(nl - 1, v=1e20)), p(lay), p(vec))
    def bb(nt=3, t=arr(3, v=500.0)):
        return lib.picaso_blackbody(ctx, ci(nt), p(t), ctypes.c_long(nw), p(arr(nw, v=1e-4)), p(arr(3, nw)))
    def dsc(ngv=ng, xi=x):
        return lib.picaso_compress_disco(ctx, ci(nw), cd(1.0), p(xi), p(arr(ng)), ci(ngv), p(arr(1)), ci(1), p(vec), p(vec))
    def sh(stream=4, dt=lay):
        return lib.picaso_get_reflected_SH(ctx, ci(nl), ci(nw), ci(ng), ci(1), p(dt), p(lev), p(lay), p(lay), p(lay), p(lay), p(lay), p(lay),
            p(lev), p(lay), p(lay), p(vec), p(u), p(u), cd(1.0), p(vec), ci(0), ci(0), ci(0), ci(1), ci(1), ci(1), cd(1.0), cd(-1.0), cd(2.0),
            cd(-0.5), cd(1.0), ci(stream), cd(0.0), ci(0), ci(0), ci(0), p(x), None)
    table = {"refl_nlevel1": lambda: refl(nlevel=1), "refl_nwno0": lambda: refl(nwno=0), "refl_numg0": lambda: refl(numg=0),
             "refl_null_dtau": lambda: refl(dtau=None), "refl_null_out": lambda: refl(out=None), "refl_bad_single_phase": lambda: refl(sp=9),
             "refl_numg_huge": lambda: refl(numg=100000), "therm_nlevel1": lambda: therm(nlevel=1), "therm_nwno_neg": lambda: therm(nwno=-5),
             "therm_null_tlevel": lambda: therm(tl=None), "therm_calc_type9": lambda: therm(ct=9), "transit_nlevel1": lambda: transit(nlevel=1),
             "transit_null": lambda: transit(z=None), "bb_ntemp0": lambda: bb(nt=0), "bb_null": lambda: bb(t=None),
             "disco_ng0": lambda: dsc(ngv=0), "disco_null": lambda: dsc(xi=None), "sh_stream3": lambda: sh(stream=3), "sh_null": lambda: sh(dt=None)}
    rc = table[name]()
    msg = lib.picaso_last_error(ctx)
    print("rc=%d %s" % (rc, (msg or b"").decode()[:110]))
    # and the context still works
    ok = refl()
    print("   next good call rc=%d" % ok)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    for c in CASES:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True, timeout=120)
        out = " | ".join(l.strip() for l in r.stdout.strip().splitlines()[-2:])
        flag = "" if (r.returncode == 0 and "rc=0 " not in out.split("|")[0] and "next good call rc=0" in out) else "   <<<<<< LOOK"
        print("%-24s exit %4d  %s%s" % (c, r.returncode, out, flag))
        if r.returncode != 0:
            print("     stderr:", r.stderr.strip().splitlines()[-1][:200] if r.stderr.strip() else "")
