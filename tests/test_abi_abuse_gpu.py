"""Bad arguments straight at the C ABI (the host-pointer forms a binding would call): every call comes back with rc != 0
and a message naming the entry point, and the NEXT good call on the same context succeeds.  Before round 5's guards a NULL
required array was a SIGSEGV (thermal / transit) or a failed hipMemcpyAsync whose sticky "last error" made every later
launch of the context report it again."""
import ctypes

import numpy as np
import pytest


@pytest.mark.gpu
def test_bad_arguments_fail_cleanly_and_do_not_poison_the_context():
    from picaso_amd import _lib
    lib, ctx = _lib.load(), _lib.context(0)
    P = ctypes.POINTER(ctypes.c_double)
    nl, nw, ng = 5, 64, 5
    ci, cd = ctypes.c_int, ctypes.c_double

    def arr(*shape, v=0.1):
        return np.full(shape, v)

    def p(a):
        return None if a is None else a.ctypes.data_as(P)
    lay, lev, vec = arr(nl - 1, nw), arr(nl, nw), arr(nw)
    u, x = arr(ng, 1, v=0.5), arr(ng, 1, nw)
    plev = arr(nl, v=1.0) * np.arange(1, nl + 1)

    def refl(nlevel=nl, nwno=nw, numg=ng, dtau=lay, out=x, sp=3):
        return lib.picaso_get_reflected_1d(
            ctx, ci(nlevel), p(vec), ci(nwno), ci(numg), ci(1), p(dtau), p(lev), p(lay), p(lay), p(lay), p(lay), p(lay), p(lay),
            p(lev), p(lay), p(lay), p(vec), p(u), p(u), cd(1.0), p(vec), ci(sp), ci(0), cd(1.0), cd(-1.0), cd(2.0), cd(-0.5),
            cd(1.0), ci(1), ci(0), ci(0), cd(0.0), p(out), None, None, None, None)

    def therm(nlevel=nl, nwno=nw, tl=arr(nl, v=500.0), ct=0):
        return lib.picaso_get_thermal_1d(
            ctx, ci(nlevel), p(arr(nw, v=1000.0)), ci(nwno), ci(ng), ci(1), p(tl), p(lay), p(lay), p(lay), p(plev), p(u), p(vec),
            ci(0), p(vec), ci(ct), p(x), None, None, None, None)

    def transit(nlevel=nl, z=arr(nl, v=1e9)):
        return lib.picaso_get_transit_1d(
            ctx, p(z), p(arr(nl, v=1e5)), ci(nlevel), ci(nw), cd(7e10), p(arr(nl - 1, v=2.3)), cd(1.38e-16), cd(1.66e-24),
            p(arr(nl - 1, v=1e5)), p(arr(nl - 1, v=500.0)), p(arr(nl - 1, v=1e20)), p(lay), p(vec))

    def bb(nt=3, t=arr(3, v=500.0)):
        return lib.picaso_blackbody(ctx, ci(nt), p(t), ctypes.c_long(nw), p(arr(nw, v=1e-4)), p(arr(3, nw)))

    def dsc(ngv=ng, xi=x):
        return lib.picaso_compress_disco(ctx, ci(nw), cd(1.0), p(xi), p(arr(ng)), ci(ngv), p(arr(1)), ci(1), p(vec), p(vec))

    def sh(stream=4, dt=lay):
        return lib.picaso_get_reflected_SH(
            ctx, ci(nl), ci(nw), ci(ng), ci(1), p(dt), p(lev), p(lay), p(lay), p(lay), p(lay), p(lay), p(lay), p(lev), p(lay),
            p(lay), p(vec), p(u), p(u), cd(1.0), p(vec), ci(0), ci(0), ci(0), ci(1), ci(1), ci(1), cd(1.0), cd(-1.0), cd(2.0),
            cd(-0.5), cd(1.0), ci(stream), cd(0.0), ci(0), ci(0), ci(0), p(x), None)

    for name, call in (("reflected", refl), ("thermal", therm), ("transit", transit), ("blackbody", bb), ("compress", dsc), ("SH", sh)):
        assert call() == 0, "%s: %s" % (name, (lib.picaso_last_error(ctx) or b"").decode())
    bad = {"refl nlevel=1": lambda: refl(nlevel=1), "refl nwno=0": lambda: refl(nwno=0), "refl numg=0": lambda: refl(numg=0),
           "refl dtau NULL": lambda: refl(dtau=None), "refl output NULL": lambda: refl(out=None),
           "refl single_phase=9": lambda: refl(sp=9), "thermal nlevel=1": lambda: therm(nlevel=1),
           "thermal nwno<0": lambda: therm(nwno=-5), "thermal tlevel NULL": lambda: therm(tl=None),
           "thermal calc_type=9": lambda: therm(ct=9), "transit nlevel=1": lambda: transit(nlevel=1),
           "transit z NULL": lambda: transit(z=None), "blackbody ntemp=0": lambda: bb(nt=0), "blackbody t NULL": lambda: bb(t=None),
           "compress ng=0": lambda: dsc(ngv=0), "compress input NULL": lambda: dsc(xi=None), "SH stream=3": lambda: sh(stream=3),
           "SH dtau NULL": lambda: sh(dt=None)}
    for name, call in bad.items():
        assert call() != 0, name
        msg = (lib.picaso_last_error(ctx) or b"").decode()
        assert len(msg) > 10, name
        assert refl() == 0, "the context is poisoned after: %s (%s)" % (name, (lib.picaso_last_error(ctx) or b"").decode())
    assert therm() == 0 and transit() == 0 and sh() == 0
