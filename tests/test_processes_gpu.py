"""The reference's callers parallelise by process (joblib / MPI ranks): several
processes share one GPU, each with its own lazily created context.  The fan-out runs in a fresh
interpreter, as in a real driver script: the parent imports the package but has not created a
context (touched HIP) when it forks."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import multiprocessing as mp, os, sys
import numpy as np
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import PLANES, rel_err
from picaso_amd import disco, fluxes, synthetic as syn      # imported before the fork, no context yet

def geometry():
    g, gw, t, tw = disco.get_angles_1d(5)
    u0, u1, ct, _, _ = disco.compute_disco(5, 1, g, t, 0.0)
    return u0, u1

def worker(seed, q):
    try:
        sc = syn.make_scene(30, 257, seed=seed)
        u0, u1 = geometry()
        x, _ = fluxes.get_reflected_1d(31, sc["wno"], 257, 5, 1, *[sc[k] for k in PLANES], 0.1, u0, u1, 1.0,
                                       np.ones(257), 3, 0, 1.0, -1.0, 2.0, -0.5, 1.0)
        q.put((seed, os.getpid(), x))
    except Exception as e:
        q.put((seed, os.getpid(), repr(e)))

if __name__ == "__main__":
    ctx = mp.get_context(METHOD)
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(s, q)) for s in (11, 12, 13)]
    for p in procs: p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0, p.exitcode
    assert len({pid for _, pid, _ in got}) == 3
    from oracle import oracle
    u0, u1 = geometry()
    for seed, _, x in got:
        assert not isinstance(x, str), x
        sc = syn.make_scene(30, 257, seed=seed)
        xo, _ = oracle.get_reflected_1d(31, sc["wno"], 257, 5, 1, *[sc[k] for k in PLANES], 0.1, u0, u1, 1.0,
                                        np.ones(257), 3, 0, 1.0, -1.0, 2.0, -0.5, 1.0)
        assert rel_err(x, xo) < 1e-8
    print("PROCESSES_OK")
'''


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["fork", "spawn"])
def test_processes_share_the_gpu(method, tmp_path):
    script = tmp_path / "fanout.py"
    script.write_text("ROOT = %r\nMETHOD = %r\n" % (ROOT, method) + SCRIPT)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PROCESSES_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
