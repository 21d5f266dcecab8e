#!/usr/bin/env python
"""A failing draw of tests/test_fuzz_gpu.py::test_fuzz_sh4_cloud_free_form (PICASO_FUZZ_OFFSET, block): the cloud-free
kernel (dtau + w0 only), the full-plane kernel on the same planes and the fp64 oracle, pairwise (run on the GPU box)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("PICASO_FUZZ_OFFSET", sys.argv[1] if len(sys.argv) > 1 else "36")
import test_fuzz_gpu as tf
from oracle import oracle
from picaso_amd import fluxes
from picaso_amd import synthetic as syn
block = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(3500 + block + 7919 * tf.OFFSET)
def rel(a, b, floor): return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))
for it in range(12):
    nlayer = int(rng.choice([1, 2, 3, 7, 19, 40, 90])); nwno = int(rng.choice([1, 5, 63, 64, 65, 130, 257, 300]))
    gs, rsc = float(10.0 ** rng.uniform(-4, 2.5)), float(10.0 ** rng.uniform(-2, 2))
    sc = syn.make_scene(nlayer, nwno, seed=2600 + 50 * block + it, stream=4, cloud=False, gas_scale=gs, ray_scale=rsc)
    ng, nt, gw, tw, u0, u1, ct = tf._geometry(rng)
    rs = rng.random(nwno) * float(rng.choice([0.0, 0.3, 1.0]))
    f0 = 1.0 + rng.random(nwno) if rng.random() < 0.5 else np.ones(nwno)
    b_top = float(rng.choice([0.0, 0.2]))
    tail = (rs, u0, u1, ct, f0, 0, 0, 0, 1, 1, 1, *tf.TTHG, 4, b_top)
    full = [sc[k] for k in ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray", "f_deltaM", "dtau_og", "tau_og", "w0_og", "cosb_og")]
    lean = [sc["dtau"], None, sc["w0"]] + [None] * 8
    xl, _ = fluxes.get_reflected_SH(nlayer + 1, nwno, ng, nt, *lean, *tail)
    xf, _ = fluxes.get_reflected_SH(nlayer + 1, nwno, ng, nt, *[np.array(a) for a in full], *tail)
    xo, _ = oracle.get_reflected_SH(nlayer + 1, nwno, ng, nt, *[np.array(a) for a in full], *tail)
    xx, _ = oracle.get_reflected_SH(nlayer + 1, nwno, ng, nt, *[np.array(a) for a in full], *tail, x80=True)
    fl = 1e-4 * np.abs(xo).max()
    print("      vs x80: clear %.1e  full %.1e  oracle %.1e" % (rel(xl, xx, fl), rel(xf, xx, fl), rel(xo, xx, fl)))
    print("it %2d nlayer %2d nwno %3d gas %.2e ray %.2e max dtau %.2e max tau %.2e  clear-oracle %.1e  full-oracle %.1e  clear-full %.1e  min w0 %.3f max w0 %.6f u0 %s"
          % (it, nlayer, nwno, gs, rsc, sc["dtau"].max(), sc["tau"].max(), rel(xl, xo, fl), rel(xf, xo, fl), rel(xl, xf, fl),
             sc["w0"].min(), sc["w0"].max(), np.round(np.ravel(u0)[:3], 3)))
