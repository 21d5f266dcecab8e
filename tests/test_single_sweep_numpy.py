"""CPU check of the ALGORITHM the HIP kernels implement: the numpy statement of the single
top-down sweep (tools/single_sweep_numpy.py) against the reference-generated golden vectors.  This
validates the invariant-imbedding reformulation (one pass, O(1) state) independently of any GPU."""
import os
import sys

import numpy as np
import pytest

from helpers import PLANES, Golden, golden_files, rel_err, scene_id

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import single_sweep_numpy as ss  # noqa: E402


@pytest.mark.parametrize("path", golden_files("scene1d_"), ids=scene_id)
def test_reflected_sweep(path):
    g = Golden(path)
    nlevel, nwno = g.inp("tau").shape
    planes = [g.inp(k) for k in PLANES]
    rs = np.zeros(nwno) + g.inp("surf_reflect")
    for case in g.cases("refl1d"):
        sp, mp, tc, lvl = (int(s[-1]) for s in case.split("_"))
        if lvl:
            continue
        x = ss.reflected_toa(nlevel, nwno, planes, rs, g.geo("ubar0").ravel(), g.geo("ubar1").ravel(),
                             g.geo("cos_theta"), g.inp("F0PI"), sp, mp, *g.tthg(), tc,
                             float(g["refl1d/%s/b_top" % case]))
        assert rel_err(x, g["refl1d/%s/xint" % case].reshape(x.shape)) < 1e-9, case


@pytest.mark.parametrize("path", golden_files("scene1d_"), ids=scene_id)
def test_thermal_sweep(path):
    g = Golden(path)
    nlevel, nwno = g.inp("tau").shape
    rs = np.zeros(nwno) + g.inp("surf_reflect")
    for hs in (0, 1):
        x = ss.thermal_toa(nlevel, g.inp("wno"), nwno, g.inp("tlevel"), g.inp("dtau_og"),
                           g.inp("w0_no_raman"), g.inp("cosb_og"), g.inp("plevel"),
                           g.geo("ubar1").ravel(), rs, hs)
        assert rel_err(x, g["therm1d/hs%d_ct0/flux" % hs].reshape(x.shape)) < 1e-8, hs
