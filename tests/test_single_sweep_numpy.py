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


@pytest.mark.parametrize("path", golden_files("scene_sh_"), ids=scene_id)
def test_sh4_block_sweep(path):
    """The 2x2-block single sweep (tools/sh_sweep_numpy.py) against the reference's SH4 results
    (LAPACK dgbsv with partial pivoting on the 11-diagonal system): no banded matrix, no pivoting
    across layers, same answer to ~1e-13 -- including thick (35-clipped) and conservative columns
    and the reference's per-angle f_deltaM compounding."""
    import sh_sweep_numpy as sh
    g = Golden(path)
    nlevel, nwno = g.inp("tau").shape
    rs = np.zeros(nwno) + g.inp("surf_reflect")
    for case in g.cases("reflsh"):
        s, f, r, sf = case.split("_")
        if s != "s4":
            continue
        wsf, wmf, psf = (int(c) for c in f[1:])
        wsr, wmr, psr = (int(c) for c in r[1:])
        x = sh.reflected_sh4(nlevel, nwno, g.inp("dtau"), g.inp("tau"), g.inp("w0"), g.inp("ftau_cld"),
                             g.inp("ftau_ray"), g.inp("f_deltaM_s4"), g.inp("dtau_og"), g.inp("tau_og"),
                             g.inp("w0_og"), g.inp("cosb_og"), rs, g.geo("ubar0").ravel(),
                             g.geo("ubar1").ravel(), g.geo("cos_theta"), g.inp("F0PI"), wsf, wmf, psf,
                             wsr, wmr, psr, *g.tthg(), 0.0, int(sf[2]))
        assert rel_err(x, g["reflsh/%s/xint" % case].reshape(x.shape)) < 1e-10, case
