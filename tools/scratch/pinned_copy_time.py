#!/usr/bin/env python
"""How long the host spends moving a finished result out of its pinned block (run on the GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from picaso_amd import _lib, device
ctx = _lib.context(0)
n = 100001
pin = device.PinnedArray((n,), ctx)
pin.array[:] = np.random.rand(n)
dst = np.empty(n)
pag = np.random.rand(n)
for name, fn in (("pinned -> numpy (copyto)", lambda: np.copyto(dst, pin.array)),
                 ("pinned -> new array (.copy())", lambda: pin.array.copy()),
                 ("pageable -> numpy (copyto)", lambda: np.copyto(dst, pag)),
                 ("np.empty(n)", lambda: np.empty(n))):
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    print("%-34s median %.1f us  min %.1f us" % (name, 1e6 * np.median(ts), 1e6 * min(ts)))
