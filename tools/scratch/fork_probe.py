"""What does a forked child see when the parent has already used the GPU?  (multiprocessing's default start method on Linux)"""
import os, sys, time, signal
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from picaso_amd import _lib, fluxes, disco
from picaso_amd import synthetic as syn
sc = syn.make_scene(20, 200, seed=2)
g, gw, t, tw = disco.get_angles_1d(5)
u0, u1, ct, _, _ = disco.compute_disco(5, 1, g, t, 0.0)
targs = (21, sc["wno"], 200, 5, 1, sc["tlevel"], sc["dtau_og"], sc["w0_no_raman"], sc["cosb_og"], sc["plevel"], u1, np.zeros(200), 0, sc["wno"] * 0, 0)
f0, _ = fluxes.get_thermal_1d(*targs, want_lvl=False)
print("parent ok", float(f0.sum()), flush=True)
pid = os.fork()
if pid == 0:
    try:
        f1, _ = fluxes.get_thermal_1d(*targs, want_lvl=False)
        print("child result equal:", bool(np.array_equal(f0, f1)), flush=True)
    except BaseException as e:
        print("child raised:", type(e).__name__, str(e)[:300], flush=True)
    os._exit(0)
t0 = time.time()
while time.time() - t0 < 30:
    r, st = os.waitpid(pid, os.WNOHANG)
    if r:
        print("child exit status", st, flush=True)
        break
    time.sleep(0.2)
else:
    os.kill(pid, signal.SIGKILL)
    print("child HUNG for 30 s: killed", flush=True)
