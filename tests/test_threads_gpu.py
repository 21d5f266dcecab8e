"""spectrum() from several Python threads on ONE opacity object returns the serial answers.  A spectrum is dozens of C
calls sharing the context's stream, arena and pool and the opacity object's workspaces; ctypes releases the GIL during
each, so unguarded threads interleave them -- measured before the guard: wrong spectra and "block still has uncollected
results".  The public entry points run under one re-entrant lock (picaso_amd/_lib.py:
CALL_LOCK); the reference itself fans out with processes (justdoit.py:4774)."""
import threading

import numpy as np
import pytest


@pytest.mark.gpu
def test_concurrent_spectra_on_one_opacity_object_equal_the_serial_ones():
    from picaso_amd import _lib
    from picaso_amd import justdoit as jdi
    from picaso_amd import optics as px
    ctx = _lib.context(0)
    nwno, nlevel = 6000, 61
    wno = np.linspace(2000.0, 33333.0, nwno)
    temps, press = [100.0, 300.0, 700.0, 1500.0, 3000.0], [1e-6, 1e-4, 1e-2, 1e-1, 1.0, 10.0, 100.0, 500.0]
    pt = [(i + 1, p, t) for i, (t, p) in enumerate((t, p) for t in temps for p in press)]
    molecular = {m: {i: 10.0 ** (-24.0 + 2.0 * np.sin(wno / 2500.0 + k) + 0.4 * np.log10(p)) for (i, p, t) in pt}
                 for k, m in enumerate(("H2O", "CH4"))}
    cia_t = [75.0, 500.0, 4000.0]
    continuum = {pr: {t: 10.0 ** (-7.0 + np.cos(wno / 4000.0 + k)) for t in cia_t} for k, pr in enumerate(("H2H2", "H2He"))}
    ray = {m: 1e-27 * (wno / 1e4) ** 4 for m in ("H2", "He")}
    opa = px.RetrieveOpacities(wno, pt, molecular, continuum, cia_t, rayleigh_opa=ray, query_method="linear", ctx=ctx)
    plev = np.logspace(-6, 2, nlevel)

    def case(k):
        prof = {"pressure": plev, "temperature": (150.0 + 1200.0 * ((np.log10(plev) + 6) / 8) ** 2) * (1 + 0.03 * k),
                "H2": np.full(nlevel, 0.84), "He": np.full(nlevel, 0.155), "H2O": np.full(nlevel, 1e-3),
                "CH4": np.full(nlevel, 5e-4)}
        c = jdi.inputs()
        c.phase_angle(0)
        c.gravity(gravity=2500.0)
        c.atmosphere(df=prof)
        c.approx(**({"raman": "none", "rt_method": "SH", "stream": 4} if k == 1 else {"raman": "none"}))
        if k >= 2:              # cloud tables: the call-by-call path next to the one-C-call path of the others
            shp = (nlevel - 1, nwno)
            opd = np.zeros(shp)
            opd[30:36] = 0.1 * (k + 1)
            c.clouds(df={"opd": opd, "w0": np.full(shp, 0.9), "g0": np.full(shp, 0.5)})
        return c

    nthreads = 4
    cases = [case(k) for k in range(nthreads)]
    serial = [c.spectrum(opa, calculation="reflected+thermal") for c in cases]
    bad, err = [0] * nthreads, [None] * nthreads

    def work(k):
        try:
            for _ in range(60):
                r = cases[k].spectrum(opa, calculation="reflected+thermal")
                if not (np.array_equal(r["albedo"], serial[k]["albedo"])
                        and np.array_equal(r["thermal"], serial[k]["thermal"])):
                    bad[k] += 1
        except Exception as e:      # noqa: BLE001  (reported below, from the main thread)
            err[k] = repr(e)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert err == [None] * nthreads, err
    assert bad == [0] * nthreads, bad


@pytest.mark.gpu
def test_forked_child_gets_a_clean_error_not_a_segfault():
    """multiprocessing's default start method on Linux: the HIP runtime does not survive fork(), a child of a process that
    has used the GPU died with SIGSEGV in its first launch.  Now the mirror says so (PicasoHipError naming the remedy)."""
    import os
    import time
    from picaso_amd import _lib, disco, fluxes
    from picaso_amd import synthetic as syn
    sc = syn.make_scene(20, 200, seed=2)
    g, gw, t, tw = disco.get_angles_1d(5)
    _, u1, _, _, _ = disco.compute_disco(5, 1, g, t, 0.0)
    targs = (21, sc["wno"], 200, 5, 1, sc["tlevel"], sc["dtau_og"], sc["w0_no_raman"], sc["cosb_og"], sc["plevel"], u1,
             np.zeros(200), 0, sc["wno"] * 0, 0)
    fluxes.get_thermal_1d(*targs, want_lvl=False)            # the parent has used the GPU
    pid = os.fork()
    if pid == 0:
        code = 3
        try:
            fluxes.get_thermal_1d(*targs, want_lvl=False)
        except _lib.PicasoHipError as e:
            code = 7 if "fork" in str(e) and "spawn" in str(e) else 4
        except BaseException:                                 # noqa: BLE001
            code = 5
        os._exit(code)
    t0 = time.time()
    status = None
    while time.time() - t0 < 60:
        r, st = os.waitpid(pid, os.WNOHANG)
        if r:
            status = st
            break
        time.sleep(0.05)
    if status is None:
        os.kill(pid, 9)
        os.waitpid(pid, 0)
        raise AssertionError("the forked child hung")
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 7, "child status %r" % (status,)
    # the parent is unharmed
    fluxes.get_thermal_1d(*targs, want_lvl=False)
