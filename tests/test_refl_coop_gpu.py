"""k_reflected_coop (csrc/toon_reflected_coop.hip: the small-launch form of get_reflected_1d, reference
fluxes.py:1009-1413) against the kernels it stands in for: the same bits for every (column, angle), whatever the
layer count (rounds of RC_R layers, ragged tails, first / last layer alone in a round), the number of disk angles
(one wave each), the geometry (zero phase or not) and where the cloud sits (the wave-uniform shortcuts fail on
different layers) -- and against the CPU oracle."""
import numpy as np
import pytest

from helpers import PLANES, rel_err

pytestmark = pytest.mark.gpu
TTHG = (1.0, -1.0, 2.0, -0.5, 1.0)


def _run(monkeypatch, coop, nlayer, nwno, ng, phase, scene_kw, rs=0.0):
    from picaso_amd import disco, fluxes
    from picaso_amd import synthetic as syn
    if coop:
        monkeypatch.delenv("PICASO_AMD_REFL_NO_COOP", raising=False)
    else:
        monkeypatch.setenv("PICASO_AMD_REFL_NO_COOP", "1")
    sc = syn.make_scene(nlayer, nwno, **scene_kw)
    if phase == 0.0:
        g, gw, t, tw = disco.get_angles_1d(ng)
        nt = 1
    else:
        g, gw, t, tw = disco.get_angles_3d(ng, 1) if False else disco.get_angles_3d(ng, 2)
        nt = 2
    u0, u1, ct, _, _ = disco.compute_disco(len(g), nt, g, t, phase)
    ct = 1.0 if phase == 0.0 else float(ct)
    f0 = np.linspace(0.8, 1.3, nwno)
    args = (nlayer + 1, sc["wno"], nwno, len(g), nt, *[sc[k] for k in PLANES], rs, u0, u1, ct, f0, 3, 0, *TTHG)
    x, _ = fluxes.get_reflected_1d(*args)
    return x, args


@pytest.mark.parametrize("nlayer", [1, 2, 3, 4, 5, 7, 8, 9, 30, 61, 90])
def test_coop_equals_grid_shapes_any_layer_count(monkeypatch, nlayer):
    kw = dict(seed=40 + nlayer)
    a, args = _run(monkeypatch, True, nlayer, 333, 5, 0.0, kw, rs=0.2)
    b, _ = _run(monkeypatch, False, nlayer, 333, 5, 0.0, kw, rs=0.2)
    assert np.array_equal(a, b)
    from oracle import oracle as orc
    xo, _ = orc.get_reflected_1d(*args)
    assert rel_err(a, xo) < 1e-8


@pytest.mark.parametrize("ng", [5, 6, 7, 8])
def test_coop_equals_grid_shapes_angle_counts(monkeypatch, ng):
    a, _ = _run(monkeypatch, True, 33, 200, ng, 0.0, dict(seed=7))
    b, _ = _run(monkeypatch, False, 33, 200, ng, 0.0, dict(seed=7))
    assert np.array_equal(a, b)


@pytest.mark.parametrize("phase", [0.6, 2.0])
def test_coop_nonzero_phase(monkeypatch, phase, oracle):
    """ubar0 != ubar1, 3 x 2 angles: the non-ZP instantiation"""
    a, args = _run(monkeypatch, True, 41, 150, 3, phase, dict(seed=9), rs=0.1)
    b, _ = _run(monkeypatch, False, 41, 150, 3, phase, dict(seed=9), rs=0.1)
    assert np.array_equal(a, b)
    xo, _ = oracle.get_reflected_1d(*args)
    assert rel_err(a, xo) < 1e-8


def test_coop_planes_that_defeat_the_shortcuts(monkeypatch, oracle):
    """Caller planes whose level optical depths are NOT the running sums of the layer depths (direct exponentials
    in every layer), a cloud in the first and in the last layer, and a cloud-free column block next to a cloudy one
    (the shortcut flags differ between the waves of one launch)."""
    from picaso_amd import disco, fluxes
    from picaso_amd import synthetic as syn
    nlayer, nwno = 23, 192
    sc = syn.make_scene(nlayer, nwno, seed=5)
    sc2 = {k: sc[k].copy() for k in PLANES}
    sc2["tau"][1:] *= 1.0 + 1e-13                     # no longer bit-exact cumulative sums
    sc2["tau_og"][3:] *= 1.0 - 1e-13
    sc2["ftau_cld"][0, :64] = 0.3                     # cloud in the first layer of the first wave only
    sc2["cosb"][0, :64] = 0.4
    sc2["cosb_og"][0, :64] = 0.5
    sc2["ftau_ray"][0, :64] = 0.7
    sc2["ftau_cld"][-1, 64:128] = 0.2                 # and in the last layer of the second
    sc2["cosb"][-1, 64:128] = 0.3
    sc2["cosb_og"][-1, 64:128] = 0.35
    sc2["ftau_ray"][-1, 64:128] = 0.8
    g, gw, t, tw = disco.get_angles_1d(5)
    u0, u1, _, _, _ = disco.compute_disco(5, 1, g, t, 0.0)
    args = (nlayer + 1, sc["wno"], nwno, 5, 1, *[sc2[k] for k in PLANES], 0.3, u0, u1, 1.0, np.ones(nwno), 3, 0, *TTHG)
    monkeypatch.delenv("PICASO_AMD_REFL_NO_COOP", raising=False)
    a, _ = fluxes.get_reflected_1d(*args)
    monkeypatch.setenv("PICASO_AMD_REFL_NO_COOP", "1")
    b, _ = fluxes.get_reflected_1d(*args)
    assert np.array_equal(a, b)
    xo, _ = oracle.get_reflected_1d(*args)
    assert rel_err(a, xo) < 1e-8


def test_coop_is_the_kernel_that_ran(monkeypatch):
    """At the 8-GPU shard size the resident call goes through k_reflected_coop (one launch, fused disk sum) and
    equals the one-angle-per-workgroup form + k_compress, albedo included."""
    from picaso_amd import _lib, device, disco, resident
    from picaso_amd import synthetic as syn
    ctx = _lib.context()
    nlayer, nwno, ng = 90, 12500, 5
    g, gw, t, tw = disco.get_angles_1d(ng)
    u0, u1, _, _, _ = disco.compute_disco(ng, 1, g, t, 0.0)
    sc = syn.make_scene(nlayer, nwno, seed=3)
    sc["F0PI"] = np.linspace(0.9, 1.1, nwno)
    sc["surf_reflect"] = np.zeros(nwno)
    d = resident.upload_scene(sc, resident.REFLECTED_PLANES + ("F0PI", "surf_reflect"), ctx=ctx)

    def run():
        x, alb = device.DeviceArray((ng, 1, nwno), ctx), device.DeviceArray((nwno,), ctx)
        resident.reflected_1d(ctx, nlayer + 1, nwno, ng, 1, d, d["surf_reflect"], u0, u1, 1.0, d["F0PI"], 3, 0, *TTHG,
                              x, gweight=gw, tweight=tw, albedo=alb)
        return x.to_host(), alb.to_host()
    monkeypatch.delenv("PICASO_AMD_REFL_NO_COOP", raising=False)
    xa, aa = run()
    monkeypatch.setenv("PICASO_AMD_REFL_NO_COOP", "1")
    xb, ab = run()
    assert np.array_equal(xa, xb) and np.array_equal(aa, ab)
    assert np.isfinite(aa).all() and aa.max() > 0


@pytest.mark.parametrize("nwno", [64, 1000, 12500])
@pytest.mark.parametrize("phase", [0.0, 0.8])
def test_coop_with_the_products_derived_plane_sets(monkeypatch, nwno, phase):
    """picaso() / the C driver leave out the planes the kernels re-derive (tau, tau_og, gcos2; for an atmosphere without
    cloud everything but dtau and w0).  At shard sizes those launches take k_reflected_coop too (round 4 sent them to the
    angle-group shapes: the product path differed from the measured one): same bits as the full plane set through the
    cooperative kernel and as the same plane set through the generic kernels."""
    from picaso_amd import _lib, device, disco, resident
    from picaso_amd import synthetic as syn
    ctx = _lib.context()
    nlayer, ng = 37, 5
    if phase == 0.0:
        g, gw, t, tw = disco.get_angles_1d(ng)
        nt, ct = 1, 1.0
    else:
        g, gw, t, tw = disco.get_angles_3d(3, 2)
        nt = 2
    u0, u1, cth, _, _ = disco.compute_disco(len(g), nt, g, t, phase)
    ct = 1.0 if phase == 0.0 else float(cth)
    for cloud in (True, False):
        sc = syn.make_scene(nlayer, nwno, seed=11, cloud=cloud)
        sc["F0PI"] = np.linspace(0.9, 1.1, nwno)
        sc["surf_reflect"] = np.full(nwno, 0.15)
        d = resident.upload_scene(sc, resident.REFLECTED_PLANES + ("F0PI", "surf_reflect"), ctx=ctx)
        keep = [k for k in resident.REFLECTED_PLANES if k not in ("tau", "tau_og", "gcos2")] if cloud else ["dtau", "w0"]
        lean = {k: d[k] for k in keep}
        assert resident.reflected_can_derive(nlayer + 1, nwno, len(g), nt, u0, u1, ct, 3, 0, 2.0)

        def run(planes):
            x, alb = device.DeviceArray((len(g), nt, nwno), ctx), device.DeviceArray((nwno,), ctx)
            resident.reflected_1d(ctx, nlayer + 1, nwno, len(g), nt, planes, d["surf_reflect"], u0, u1, ct, d["F0PI"], 3, 0,
                                  *TTHG, x, gweight=gw, tweight=tw, albedo=alb)
            return x.to_host(), alb.to_host()
        monkeypatch.delenv("PICASO_AMD_REFL_NO_COOP", raising=False)
        full_c, lean_c = run(d), run(lean)
        monkeypatch.setenv("PICASO_AMD_REFL_NO_COOP", "1")
        lean_g = run(lean)
        for a, b in ((full_c, lean_c), (lean_c, lean_g)):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (cloud, nwno, phase)
        assert np.isfinite(lean_c[1]).all() and lean_c[1].max() > 0
