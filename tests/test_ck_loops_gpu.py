"""The correlated-k Gauss-point loop of the reference's picaso() around the SH solvers and around the 3-D solvers
(justdoit.py:256-307, 364-380, 488-516) on the GPU (csrc/ckloop.hip), against tests/golden/ck_rt.npz -- outputs of the
reference's own loop (make_golden.py ck_rt) -- at 1e-9, and bit for bit against the per-Gauss-point GPU calls; plus the SH
option corner of tests/golden/sh_extra_*.npz (form index 2 'isotropic' on each form argument, b_top != 0)."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, PLANES, Golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-9
TTHG = (1.0, -1.0, 2.0, -0.5, 1.0)
CK_NAMES = ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray", "gcos2", "dtau_og", "tau_og", "w0_og", "cosb_og",
            "w0_no_raman", "f_deltaM")


def _sh_case(case):
    s, f, r, sf = case.split("_")
    return int(s[1]), [int(c) for c in f[1:]], [int(c) for c in r[1:]], int(sf[2])


@pytest.fixture(scope="module")
def fx():
    return (np.load(os.path.join(GOLDEN, "ck_rt.npz")), np.load(os.path.join(GOLDEN, "ck.npz")),
            np.load(os.path.join(GOLDEN, "optics.npz")))


@pytest.mark.parametrize("gname", ["g5", "g3x2"])
@pytest.mark.parametrize("stream", [2, 4])
def test_sh_inside_the_gauss_loop(fx, gname, stream):
    from picaso_amd import _lib, resident
    from picaso_amd.device import DeviceArray
    r, ck, og = fx
    ctx = _lib.context()
    wno, wts = og["in/wno"], ck["in/gauss_wts"]
    nwno, ngauss, nlevel = wno.size, wts.size, r["sh/tlevel"].size
    geo = {k: r["sh/%s/geo/%s" % (gname, k)] for k in ("numg", "numt", "ubar0", "ubar1", "cos_theta", "gweight", "tweight")}
    ng, nt, ct = int(geo["numg"]), int(geo["numt"]), float(geo["cos_theta"])
    pl = {nm: np.ascontiguousarray(ck["de1_s%d/%s" % (stream, nm)]) for nm in CK_NAMES}
    d = {nm: DeviceArray.from_host(pl[nm], ctx) for nm in CK_NAMES}
    d_rs, d_f0 = DeviceArray.from_host(r["sh/surf_reflect"], ctx), DeviceArray.from_host(r["sh/F0PI"], ctx)
    d_wno = DeviceArray.from_host(wno, ctx)
    keys = [k[:-len("/xint_at_top")] for k in r.files if k.startswith("sh/%s/s%d_" % (gname, stream))
            and k.endswith("/xint_at_top")]
    assert len(keys) == 3
    for key in keys:
        _, forms, rays, sf = _sh_case(key.split("/")[-1])
        xint, alb = DeviceArray((ng, nt, nwno), ctx), DeviceArray((nwno,), ctx)
        resident.reflected_SH_ck(ctx, nlevel, nwno, ngauss, ng, nt, d, d_rs, geo["ubar0"], geo["ubar1"], ct, d_f0, *forms,
                                 *rays, *TTHG, stream, wts, xint, single_form=sf, gweight=geo["gweight"],
                                 tweight=geo["tweight"], albedo=alb)
        got = xint.to_host()
        assert rel_err(got, r[key + "/xint_at_top"]) < TOL, key
        assert rel_err(alb.to_host(), r[key + "/albedo"]) < TOL, key
        # the reference's loop with the monochromatic GPU call in the solver's place: the same bits
        acc = 0
        for ig in range(ngauss):
            d1 = {nm: DeviceArray.from_host(np.ascontiguousarray(pl[nm][:, :, ig]), ctx) for nm in resident.SH_PLANES}
            x1 = DeviceArray((ng, nt, nwno), ctx)
            resident.reflected_SH(ctx, nlevel, nwno, ng, nt, d1, d_rs, geo["ubar0"], geo["ubar1"], ct, d_f0, *forms, *rays,
                                  *TTHG, stream, x1, single_form=sf)
            acc = acc + x1.to_host() * wts[ig]
        assert np.array_equal(got, acc), key
    for hs in (0, 1):
        key = "sh/%s/thermal_s%d_hs%d" % (gname, stream, hs)
        flux, disk = DeviceArray((ng, nt, nwno), ctx), DeviceArray((nwno,), ctx)
        resident.thermal_SH_ck(ctx, nlevel, d_wno, nwno, ngauss, ng, nt, r["sh/tlevel"], d["dtau"], d["w0"], d["cosb_og"],
                               r["sh/plevel"], geo["ubar1"], d_rs, stream, hs, True, wts, flux, tau=d["tau"],
                               gweight=geo["gweight"], tweight=geo["tweight"], flux_disk=disk)
        assert rel_err(flux.to_host(), r[key + "/flux_at_top"]) < TOL, key
        assert rel_err(disk.to_host(), r[key + "/thermal"]) < TOL, key


def test_sh_gauss_loop_with_one_point_is_the_plain_call():
    from picaso_amd import _lib, resident
    from picaso_amd import synthetic as syn
    from picaso_amd.device import DeviceArray
    ctx = _lib.context()
    nlayer, nwno = 21, 333
    sc = syn.make_scene(nlayer, nwno, seed=12, stream=4)
    d = {k: DeviceArray.from_host(sc[k], ctx) for k in resident.SH_PLANES}
    rs, f0 = DeviceArray.from_host(np.full(nwno, 0.15), ctx), DeviceArray.from_host(np.linspace(0.9, 1.1, nwno), ctx)
    u = np.array([[0.9], [0.6], [0.3]])
    a, b = DeviceArray((3, 1, nwno), ctx), DeviceArray((3, 1, nwno), ctx)
    opts = (0, 0, 0, 1, 1, 1, *TTHG, 4)
    resident.reflected_SH_ck(ctx, nlayer + 1, nwno, 1, 3, 1, d, rs, u, u, 1.0, f0, *opts, np.ones(1), a)
    resident.reflected_SH(ctx, nlayer + 1, nwno, 3, 1, d, rs, u, u, 1.0, f0, *opts, b)
    assert np.array_equal(a.to_host(), b.to_host())
    with pytest.raises(Exception, match="ngauss"):
        resident.reflected_SH_ck(ctx, nlayer + 1, nwno, 33, 3, 1, d, rs, u, u, 1.0, f0, *opts, np.ones(33), a)


def _facet_major(a):
    """(rows, nwno, ng, nt, ngauss) as the reference holds it -> (ng*nt, rows, nwno, ngauss)"""
    rows, nwno, ng, nt, nk = a.shape
    return np.ascontiguousarray(np.moveaxis(a.reshape(rows, nwno, ng * nt, nk), 2, 0))


def test_3d_inside_the_gauss_loop(fx):
    from picaso_amd import _lib, resident
    from picaso_amd.device import DeviceArray
    r, _, _ = fx
    ctx = _lib.context()
    fam = "r3d"
    wts = r[fam + "/in/gauss_wts"]
    geo = {k: r["%s/geo/%s" % (fam, k)] for k in ("numg", "numt", "ubar0", "ubar1", "cos_theta", "gweight", "tweight")}
    ng, nt, ct = int(geo["numg"]), int(geo["numt"]), float(geo["cos_theta"])
    nlevel, nwno = r[fam + "/in/tau"].shape[:2]
    ngauss = wts.size
    pl = {k: r["%s/in/%s" % (fam, k)] for k in PLANES + ("w0_no_raman",)}
    d = {k: DeviceArray.from_host(_facet_major(v), ctx) for k, v in pl.items()}
    d_rs, d_f0 = DeviceArray.from_host(r[fam + "/in/surf_reflect"], ctx), DeviceArray.from_host(r[fam + "/in/F0PI"], ctx)
    d_wno = DeviceArray.from_host(r[fam + "/in/wno"], ctx)
    for sp, mp in ((3, 0), (0, 1), (1, 0)):
        key = "%s/refl_sp%d_mp%d" % (fam, sp, mp)
        xint, alb = DeviceArray((ng, nt, nwno), ctx), DeviceArray((nwno,), ctx)
        resident.reflected_3d_ck(ctx, nlevel, nwno, ngauss, ng, nt, d, d_rs, geo["ubar0"], geo["ubar1"], ct, d_f0, sp, mp,
                                 *TTHG, wts, xint, gweight=geo["gweight"], tweight=geo["tweight"], albedo=alb)
        got = xint.to_host()
        assert rel_err(got, r[key + "/xint_at_top"]) < TOL, key
        assert rel_err(alb.to_host(), r[key + "/albedo"]) < TOL, key
        # per Gauss point through the facet-fastest 3-D call on the reference's own slices: the same bits
        acc = 0
        for ig in range(ngauss):
            d1 = {k: DeviceArray.from_host(np.ascontiguousarray(pl[k][..., ig]), ctx) for k in PLANES}
            x1 = DeviceArray((ng, nt, nwno), ctx)
            resident.reflected_3d(ctx, nlevel, nwno, ng, nt, d1, d_rs, geo["ubar0"], geo["ubar1"], ct, d_f0, sp, mp, *TTHG, x1)
            acc = acc + x1.to_host() * wts[ig]
        assert np.array_equal(got, acc), key
        if (sp, mp) == (3, 0):      # derived planes left out (tau, tau_og, gcos2): the same bits again
            lean = {k: v for k, v in d.items() if k not in ("tau", "tau_og", "gcos2")}
            x2 = DeviceArray((ng, nt, nwno), ctx)
            resident.reflected_3d_ck(ctx, nlevel, nwno, ngauss, ng, nt, lean, d_rs, geo["ubar0"], geo["ubar1"], ct, d_f0, sp,
                                     mp, *TTHG, wts, x2)
            assert np.array_equal(x2.to_host(), got)
    for hs in (0, 1):
        key = "%s/therm_hs%d" % (fam, hs)
        flux, disk = DeviceArray((ng, nt, nwno), ctx), DeviceArray((nwno,), ctx)
        resident.thermal_3d_ck(ctx, nlevel, d_wno, nwno, ngauss, ng, nt, r[fam + "/in/tlevel"], d["dtau_og"], d["w0_no_raman"],
                               d["cosb_og"], r[fam + "/in/plevel"], geo["ubar1"], d_rs, hs, wts, flux,
                               gweight=geo["gweight"], tweight=geo["tweight"], flux_disk=disk)
        got = flux.to_host()
        assert rel_err(got, r[key + "/flux_at_top"]) < TOL, key
        assert rel_err(disk.to_host(), r[key + "/thermal"]) < TOL, key
        acc = 0
        for ig in range(ngauss):
            sl = [DeviceArray.from_host(np.ascontiguousarray(pl[k][..., ig]), ctx) for k in ("dtau_og", "w0_no_raman", "cosb_og")]
            f1 = DeviceArray((ng, nt, nwno), ctx)
            resident.thermal_3d(ctx, nlevel, d_wno, nwno, ng, nt, r[fam + "/in/tlevel"], *sl, r[fam + "/in/plevel"],
                                geo["ubar1"], d_rs, hs, f1)
            acc = acc + f1.to_host() * wts[ig]
        assert np.array_equal(got, acc), key


@pytest.mark.parametrize("name", ["cfg3like", "phase60"])
def test_sh_isotropic_form_and_b_top(name):
    """Form index 2 on each of the three form arguments and b_top != 0 (tests/golden/sh_extra_<name>.npz)."""
    from picaso_amd import fluxes
    g = Golden(os.path.join(GOLDEN, "scene_sh_%s.npz" % name))
    x = Golden(os.path.join(GOLDEN, "sh_extra_%s.npz" % name))
    nlevel, nwno = g.inp("tau").shape

    def run(stream, forms, rays, sf, b_top):
        return fluxes.get_reflected_SH(
            nlevel, nwno, g.geo("numg"), g.geo("numt"), g.inp("dtau"), g.inp("tau"), g.inp("w0"), g.inp("cosb"),
            g.inp("ftau_cld"), g.inp("ftau_ray"), g.inp("f_deltaM_s%d" % stream).copy(), g.inp("dtau_og"), g.inp("tau_og"),
            g.inp("w0_og"), g.inp("cosb_og"), g.inp("surf_reflect"), g.geo("ubar0"), g.geo("ubar1"), g.geo("cos_theta"),
            g.inp("F0PI"), *forms, *rays, *g.tthg(), stream, b_top=b_top, flx=0, single_form=sf)[0]
    cases = x.cases("reflsh")
    assert len(cases) == 14
    for case in cases:
        stream, forms, rays, sf = _sh_case(case)
        assert rel_err(run(stream, forms, rays, sf, 0.0), x["reflsh/%s/xint" % case]) < TOL, case
    for case in x.cases("btop"):
        stream, forms, rays, sf = _sh_case(case)
        b_top = float(x["btop/%s/b_top" % case])
        assert rel_err(run(stream, forms, rays, sf, b_top), x["btop/%s/xint" % case]) < TOL, case
