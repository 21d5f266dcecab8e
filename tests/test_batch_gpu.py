"""Batched launches (picaso_get_reflected_1d_batch_dev / picaso_get_thermal_1d_batch_dev: B spectra of one
shape in ONE grid, SURVEY 8(f) rank 4 -- the reference runs them as separate processes, driver.py:405-426,
justdoit.py:4741-4777, each through get_reflected_1d / get_thermal_1d, fluxes.py:1009-1413 / 1682-1912).

The contract: spectrum s of a batch is bit-identical (np.array_equal) to the single call on its own arguments, for
every launch shape the batch may take (fused five-angle waves, one wave per SIMD, angle groups, non-zero phase,
generic options), for B distinct atmospheres, for one atmosphere under B geometries, and for mixtures; plus a
sample of every batch against the CPU oracle."""
import numpy as np
import pytest

from helpers import PLANES, rel_err

pytestmark = pytest.mark.gpu
TTHG = (1.0, -1.0, 2.0, -0.5, 1.0)


def _geom(ng, phase):
    from picaso_amd import disco
    if phase == 0.0:
        g, gw, t, tw = disco.get_angles_1d(ng)
        nt = 1
    else:
        g, gw, t, tw = disco.get_angles_3d(ng, 2)
        nt = 2
    u0, u1, ct, _, _ = disco.compute_disco(len(g), nt, g, t, phase)
    return len(g), nt, u0, u1, (1.0 if phase == 0.0 else float(ct)), gw, tw


def _scenes(B, nlayer, nwno, seed0, **kw):
    from picaso_amd import _lib, resident
    from picaso_amd import synthetic as syn
    ctx = _lib.context()
    host, dev = [], []
    for s in range(B):
        sc = syn.make_scene(nlayer, nwno, seed=seed0 + s, **kw)
        sc["F0PI"] = np.linspace(0.8 + 0.05 * s, 1.3, nwno)
        sc["surf_reflect"] = np.full(nwno, 0.1 * (s % 3))
        host.append(sc)
        dev.append(resident.upload_scene(sc, resident.REFLECTED_PLANES + ("F0PI", "surf_reflect", "w0_no_raman"), ctx=ctx))
    return ctx, host, dev


def _single_reflected(ctx, d, nlayer, nwno, geom, opts, fuse=True):
    from picaso_amd import device, resident
    ng, nt, u0, u1, ct, gw, tw = geom
    x, alb = device.DeviceArray((ng, nt, nwno), ctx), device.DeviceArray((nwno,), ctx)
    resident.reflected_1d(ctx, nlayer + 1, nwno, ng, nt, d, d["surf_reflect"], u0, u1, ct, d["F0PI"], *opts, x,
                          gweight=gw if fuse else None, tweight=tw if fuse else None, albedo=alb if fuse else None)
    return x.to_host(), (alb.to_host() if fuse else None)


def _batch_reflected(ctx, devs, nlayer, nwno, geoms, opts, fuse=True):
    from picaso_amd import device, resident
    B = len(devs)
    ng, nt = geoms[0][0], geoms[0][1]
    xs = [device.DeviceArray((ng, nt, nwno), ctx) for _ in range(B)]
    als = [device.DeviceArray((nwno,), ctx) for _ in range(B)]
    same = all(g is geoms[0] for g in geoms)
    u0 = geoms[0][2] if same else np.stack([g[2] for g in geoms])
    u1 = geoms[0][3] if same else np.stack([g[3] for g in geoms])
    ct = geoms[0][4] if same else np.array([g[4] for g in geoms])
    resident.reflected_1d_batch(ctx, nlayer + 1, nwno, ng, nt, devs, [d["surf_reflect"] for d in devs], u0, u1, ct,
                                [d["F0PI"] for d in devs], *opts, xs, gweight=geoms[0][5] if fuse else None,
                                tweight=geoms[0][6] if fuse else None, albedo=als if fuse else None)
    return [x.to_host() for x in xs], [a.to_host() for a in als] if fuse else None


@pytest.mark.parametrize("B", [2, 4, 7])
@pytest.mark.parametrize("nwno", [300, 12500, 33000])
def test_reflected_batch_equals_single_calls(B, nwno, oracle):
    """B distinct atmospheres, default options, zero phase: 300 columns run as angle groups, 12 500 x B crosses from
    one wave per SIMD (B = 2: the all-register kernel) to two, 33 000 x B is the throughput regime."""
    nlayer = 90 if nwno > 1000 else 37
    ctx, host, devs = _scenes(B, nlayer, nwno, 100 + B)
    geom = _geom(5, 0.0)
    opts = (3, 0, *TTHG)
    xb, ab = _batch_reflected(ctx, devs, nlayer, nwno, [geom] * B, opts)
    for s in range(B):
        x1, a1 = _single_reflected(ctx, devs[s], nlayer, nwno, geom, opts)
        assert np.array_equal(xb[s], x1), "spectrum %d of %d differs from its single call" % (s, B)
        assert np.array_equal(ab[s], a1)
    # oracle on a column sample of the last spectrum
    sc = host[-1]
    idx = np.linspace(0, nwno - 1, min(nwno, 160)).astype(int)
    args = (nlayer + 1, sc["wno"][idx], idx.size, 5, 1, *[np.ascontiguousarray(sc[k][:, idx]) for k in PLANES],
            sc["surf_reflect"][idx], geom[2], geom[3], 1.0, sc["F0PI"][idx], 3, 0, *TTHG)
    xo, _ = oracle.get_reflected_1d(*args)
    assert rel_err(xb[-1][:, :, idx], xo) < 1e-9


@pytest.mark.parametrize("opts", [(3, 0, *TTHG), (0, 1, 0.9, -0.8, 2.0, -0.4, 0.9), (2, 0, *TTHG)])
def test_reflected_batch_one_atmosphere_several_geometries(opts, oracle):
    """The phase-curve form: ONE plane set, B geometries (ubar0 != ubar1, cos_theta per spectrum): the shared-plane
    workgroup order, the non-zero-phase kernels, generic options."""
    nlayer, nwno, B = 41, 2100, 4
    ctx, host, devs = _scenes(1, nlayer, nwno, 7)
    geoms = [_geom(3, ph) for ph in (0.3, 0.9, 1.6, 2.2)]
    xb, ab = _batch_reflected(ctx, [devs[0]] * B, nlayer, nwno, geoms, opts)
    for s in range(B):
        x1, a1 = _single_reflected(ctx, devs[0], nlayer, nwno, geoms[s], opts)
        assert np.array_equal(xb[s], x1)
        assert np.array_equal(ab[s], a1)
    sc, g = host[0], geoms[2]
    args = (nlayer + 1, sc["wno"], nwno, g[0], g[1], *[sc[k] for k in PLANES], sc["surf_reflect"], g[2], g[3], g[4],
            sc["F0PI"], opts[0], opts[1], *opts[2:])
    xo, _ = oracle.get_reflected_1d(*args)
    assert rel_err(xb[2], xo) < 1e-8


def test_reflected_batch_mixed_zero_and_nonzero_phase():
    """A batch whose geometries are not all symmetric runs the general-geometry kernel for every spectrum; the
    zero-phase member must still equal its single call (which takes the symmetric-geometry kernel)."""
    nlayer, nwno = 30, 5000
    ctx, host, devs = _scenes(3, nlayer, nwno, 21)
    geoms = [_geom(3, 0.7), _geom(3, 0.7), _geom(3, 0.7)]
    # make spectrum 1 symmetric: ubar0 := ubar1, cos_theta stays what it is (not 1: the generic-cos_theta path)
    g1 = list(geoms[1])
    g1[2] = g1[3].copy()
    geoms[1] = tuple(g1)
    opts = (3, 0, *TTHG)
    xb, ab = _batch_reflected(ctx, devs, nlayer, nwno, geoms, opts)
    for s in range(3):
        x1, a1 = _single_reflected(ctx, devs[s], nlayer, nwno, geoms[s], opts)
        assert np.array_equal(xb[s], x1), s
        assert np.array_equal(ab[s], a1), s


@pytest.mark.parametrize("ng", [6, 8])
def test_reflected_batch_more_than_five_angles(ng):
    """6 and 8 disk angles: two launch chunks per batch, the disk sum continued across them per spectrum."""
    nlayer, nwno, B = 25, 40000, 2
    ctx, host, devs = _scenes(B, nlayer, nwno, 33)
    geom = _geom(ng, 0.0)
    opts = (3, 0, *TTHG)
    xb, ab = _batch_reflected(ctx, devs, nlayer, nwno, [geom] * B, opts)
    for s in range(B):
        x1, a1 = _single_reflected(ctx, devs[s], nlayer, nwno, geom, opts)
        assert np.array_equal(xb[s], x1) and np.array_equal(ab[s], a1)


def test_reflected_batch_without_the_fused_disk_sum_and_bad_arguments():
    from picaso_amd import _lib, device, resident
    nlayer, nwno = 12, 700
    ctx, host, devs = _scenes(2, nlayer, nwno, 3)
    geom = _geom(5, 0.0)
    opts = (3, 0, *TTHG)
    xb, _ = _batch_reflected(ctx, devs, nlayer, nwno, [geom] * 2, opts, fuse=False)
    for s in range(2):
        x1, _ = _single_reflected(ctx, devs[s], nlayer, nwno, geom, opts, fuse=False)
        assert np.array_equal(xb[s], x1)
    with pytest.raises(_lib.PicasoHipError, match="single_phase"):
        _batch_reflected(ctx, devs, nlayer, nwno, [geom] * 2, (9, 0, *TTHG))
    with pytest.raises(Exception, match="per-spectrum"):
        x = [device.DeviceArray((5, 1, nwno), ctx)]
        resident.reflected_1d_batch(ctx, nlayer + 1, nwno, 5, 1, devs, devs[0]["surf_reflect"], geom[2], geom[3], 1.0,
                                    devs[0]["F0PI"], *opts, x)


def _single_thermal(ctx, sc, d, wno_d, nlayer, nwno, geom, hard):
    from picaso_amd import device, resident
    ng, nt, u0, u1, ct, gw, tw = geom
    f, disk = device.DeviceArray((ng, nt, nwno), ctx), device.DeviceArray((nwno,), ctx)
    resident.thermal_1d(ctx, nlayer + 1, wno_d, nwno, ng, nt, sc["tlevel"], d["dtau_og"], d["w0_no_raman"], d["cosb_og"],
                        sc["plevel"], u1, d["surf_reflect"], hard, f, gweight=gw, tweight=tw, flux_disk=disk)
    return f.to_host(), disk.to_host()


@pytest.mark.parametrize("B", [2, 4, 7])
@pytest.mark.parametrize("nwno", [400, 10000, 30000])
def test_thermal_batch_equals_single_calls(B, nwno, oracle):
    """B atmospheres with their own level temperatures; 1e4 columns is BASELINE configs[1]'s size (the single call
    takes the cooperative kernel there, the batch the lane-per-column one: same bits)."""
    from picaso_amd import device, resident
    nlayer = 90 if nwno > 1000 else 33
    ctx, host, devs = _scenes(B, nlayer, nwno, 50 + B)
    for s, sc in enumerate(host):
        sc["tlevel"] = sc["tlevel"] * (1.0 + 0.03 * s)
    geom = _geom(5, 0.0)
    ng, nt, u0, u1, ct, gw, tw = geom
    wno_d = device.DeviceArray.from_host(host[0]["wno"], ctx)
    fs = [device.DeviceArray((ng, nt, nwno), ctx) for _ in range(B)]
    ds = [device.DeviceArray((nwno,), ctx) for _ in range(B)]
    resident.thermal_1d_batch(ctx, nlayer + 1, wno_d, nwno, ng, nt, np.stack([sc["tlevel"] for sc in host]),
                              [d["dtau_og"] for d in devs], [d["w0_no_raman"] for d in devs],
                              [d["cosb_og"] for d in devs], np.stack([sc["plevel"] for sc in host]), u1,
                              [d["surf_reflect"] for d in devs], 0, fs, gweight=gw, tweight=tw, flux_disk=ds)
    for s in range(B):
        f1, d1 = _single_thermal(ctx, host[s], devs[s], wno_d, nlayer, nwno, geom, 0)
        assert np.array_equal(fs[s].to_host(), f1), s
        assert np.array_equal(ds[s].to_host(), d1), s
    sc = host[-1]
    idx = np.linspace(0, nwno - 1, min(nwno, 128)).astype(int)
    targs = (nlayer + 1, sc["wno"][idx], idx.size, 5, 1, sc["tlevel"], np.ascontiguousarray(sc["dtau_og"][:, idx]),
             np.ascontiguousarray(sc["w0_no_raman"][:, idx]), np.ascontiguousarray(sc["cosb_og"][:, idx]), sc["plevel"],
             u1, sc["surf_reflect"][idx], 0, sc["wno"][idx] * 0, 0)
    fo, _ = oracle.get_thermal_1d(*targs)
    assert rel_err(fs[-1].to_host()[:, :, idx], fo) < 1e-9


def test_thermal_batch_several_geometries_hard_surface():
    from picaso_amd import device, resident
    nlayer, nwno, B = 20, 3000, 3
    ctx, host, devs = _scenes(B, nlayer, nwno, 77)
    geoms = [_geom(3, ph) for ph in (0.2, 1.0, 1.9)]
    ng, nt = geoms[0][0], geoms[0][1]
    wno_d = device.DeviceArray.from_host(host[0]["wno"], ctx)
    fs = [device.DeviceArray((ng, nt, nwno), ctx) for _ in range(B)]
    ds = [device.DeviceArray((nwno,), ctx) for _ in range(B)]
    resident.thermal_1d_batch(ctx, nlayer + 1, wno_d, nwno, ng, nt, np.stack([sc["tlevel"] for sc in host]),
                              [d["dtau_og"] for d in devs], [d["w0_no_raman"] for d in devs],
                              [d["cosb_og"] for d in devs], np.stack([sc["plevel"] for sc in host]),
                              np.stack([g[3] for g in geoms]), [d["surf_reflect"] for d in devs], 1, fs,
                              gweight=geoms[0][5], tweight=geoms[0][6], flux_disk=ds)
    for s in range(B):
        f1, d1 = _single_thermal(ctx, host[s], devs[s], wno_d, nlayer, nwno, geoms[s], 1)
        assert np.array_equal(fs[s].to_host(), f1), s
        assert np.array_equal(ds[s].to_host(), d1), s


# ---- the product call: justdoit.spectrum_batch ----
import os  # noqa: E402

from helpers import GOLDEN  # noqa: E402
from test_devices_gpu import _same  # noqa: E402

DB = os.path.join(GOLDEN, "synthetic_opacities.db")


def _product_case(og, jdi, k, cloud=True, phase=0.0, approx=None):
    """Atmosphere k of a 'retrieval': its own temperature offset, abundances, cloud depth and surface."""
    case = jdi.inputs()
    if phase == 0.0:
        case.phase_angle(0)
    else:
        case.phase_angle(phase, num_gangle=4, num_tangle=2)      # 8 facets: a geometry of its own
    case.gravity(gravity=float(og["in/gravity"]) * (1.0 + 0.05 * k), radius=7.1e9, mass=1.9e30)
    prof = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"] * (1.0 + 0.02 * k)}
    for j, m in enumerate(("H2", "He", "H2O", "CH4")):
        prof[m] = og["in/mix/" + m] * (1.0 + (0.1 * k if j >= 2 else 0.0))
    case.atmosphere(df=prof)
    if cloud:
        case.clouds(df={"opd": og["in/cld_opd"] * (0.5 + 0.5 * k), "w0": og["in/cld_w0"], "g0": og["in/cld_g0"]})
    nwno = len(og["in/wno"])
    case.star(relative_flux=1.0 + 0.3 * np.sin(np.arange(nwno) / 7.0), radius=6.9e10, semi_major=7.5e12)
    case.surface_reflect(0.05 * k)
    case.approx(**(approx or dict(raman="none", delta_eddington=True)))
    return case


@pytest.mark.parametrize("calc", ["reflected", "thermal", "reflected+thermal", "reflected+thermal+transmission"])
@pytest.mark.parametrize("B", [2, 4, 7])
def test_spectrum_batch_equals_spectrum_calls(calc, B):
    """B atmospheres through spectrum_batch (one solver launch per leg for the whole batch) against B spectrum()
    calls: every key of every output dictionary, full_output included, bit for bit."""
    from picaso_amd import justdoit as jdi
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    want = [_product_case(og, jdi, k).spectrum(opa, calculation=calc, full_output=True) for k in range(B)]
    got = jdi.spectrum_batch([_product_case(og, jdi, k) for k in range(B)], opa, calculation=calc, full_output=True)
    assert len(got) == B
    for w, g in zip(want, got):
        _same(w, g)


def test_spectrum_batch_mixed_cases_and_chunks():
    """Cloudy and cloud-free atmospheres, two phase angles and a non-default approximation in one call, chunks of 3:
    cases that cannot share a launch go alone, the outputs still come back in order and equal spectrum()'s."""
    from picaso_amd import justdoit as jdi
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    opa = jdi.opannection(filename_db=DB, query_method="linear")

    def cases():
        return [_product_case(og, jdi, 0), _product_case(og, jdi, 1, cloud=False), _product_case(og, jdi, 2, phase=0.8),
                _product_case(og, jdi, 3, approx=dict(raman="none", single_phase="OTHG", multi_phase="N=1")),
                _product_case(og, jdi, 4, cloud=False), _product_case(og, jdi, 5, phase=0.8),
                _product_case(og, jdi, 6, approx=dict(raman="pollack")) if getattr(opa, "raman_db", None) is not None
                else _product_case(og, jdi, 6)]
    want = [c.spectrum(opa, calculation="reflected+thermal") for c in cases()]
    got = jdi.spectrum_batch(cases(), opa, calculation="reflected+thermal", batch_size=3)
    for w, g in zip(want, got):
        _same(w, g)
    with pytest.raises(Exception, match="gravity"):
        bad = jdi.inputs()
        bad.phase_angle(0)
        bad.atmosphere(df={"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"], "H2": og["in/mix/H2"]})
        jdi.spectrum_batch([bad], opa)


# ---- spherical harmonics ----
@pytest.mark.parametrize("stream", [2, 4])
@pytest.mark.parametrize("B,nwno", [(2, 500), (4, 12500), (7, 3000)])
def test_reflected_SH_batch_equals_single_calls(stream, B, nwno, oracle):
    """picaso_get_reflected_SH_batch_dev (fluxes.py:2675-2976 per spectrum): B atmospheres, default SH options; 4 x
    12 500 columns is the configs[3] shard size and crosses into the XCD-ordered grid."""
    from picaso_amd import _lib, device, resident
    from picaso_amd import synthetic as syn
    ctx = _lib.context()
    nlayer = 90 if nwno > 5000 else 31
    geom = _geom(5, 0.0)
    ng, nt, u0, u1, ct, gw, tw = geom
    opts = (0, 0, 0, 1, 1, 1)
    host, devs = [], []
    for s in range(B):
        sc = syn.make_scene(nlayer, nwno, seed=300 + 10 * stream + s, stream=stream)
        sc["F0PI"] = np.linspace(0.9, 1.2 + 0.1 * s, nwno)
        sc["surf_reflect"] = np.full(nwno, 0.05 * s)
        host.append(sc)
        devs.append(resident.upload_scene(sc, resident.SH_PLANES + ("F0PI", "surf_reflect"), ctx=ctx))
    xs = [device.DeviceArray((ng, nt, nwno), ctx) for _ in range(B)]
    als = [device.DeviceArray((nwno,), ctx) for _ in range(B)]
    resident.reflected_SH_batch(ctx, nlayer + 1, nwno, ng, nt, devs, [d["surf_reflect"] for d in devs], u0, u1, ct,
                                [d["F0PI"] for d in devs], *opts, *TTHG, stream, xs, gweight=gw, tweight=tw, albedo=als)
    for s in range(B):
        x1, a1 = device.DeviceArray((ng, nt, nwno), ctx), device.DeviceArray((nwno,), ctx)
        resident.reflected_SH(ctx, nlayer + 1, nwno, ng, nt, devs[s], devs[s]["surf_reflect"], u0, u1, ct,
                              devs[s]["F0PI"], *opts, *TTHG, stream, x1, gweight=gw, tweight=tw, albedo=a1)
        assert np.array_equal(xs[s].to_host(), x1.to_host()), s
        assert np.array_equal(als[s].to_host(), a1.to_host()), s
    sc = host[-1]
    idx = np.linspace(0, nwno - 1, 96).astype(int)
    planes = [np.array(sc[k][:, idx], order="C") for k in resident.SH_PLANES]
    xo, _ = oracle.get_reflected_SH(nlayer + 1, idx.size, ng, nt, *planes, sc["surf_reflect"][idx], u0, u1, ct,
                                    sc["F0PI"][idx], *opts, *TTHG, stream)
    assert rel_err(xs[-1].to_host()[:, :, idx], xo) < 1e-9


def test_reflected_SH_batch_geometries_and_generic_options():
    """One plane set under three geometries (ubar0 != ubar1), OTHG weights: the generic kernel, per-spectrum angles."""
    from picaso_amd import _lib, device, resident
    from picaso_amd import synthetic as syn
    ctx = _lib.context()
    nlayer, nwno, B = 27, 1500, 3
    sc = syn.make_scene(nlayer, nwno, seed=77, stream=4)
    sc["F0PI"], sc["surf_reflect"] = np.ones(nwno), np.full(nwno, 0.2)
    d = resident.upload_scene(sc, resident.SH_PLANES + ("F0PI", "surf_reflect"), ctx=ctx)
    geoms = [_geom(3, ph) for ph in (0.4, 1.1, 2.0)]
    ng, nt = geoms[0][0], geoms[0][1]
    opts = (1, 1, 1, 1, 0, 1)
    xs = [device.DeviceArray((ng, nt, nwno), ctx) for _ in range(B)]
    resident.reflected_SH_batch(ctx, nlayer + 1, nwno, ng, nt, [d] * B, d["surf_reflect"], np.stack([g[2] for g in geoms]),
                                np.stack([g[3] for g in geoms]), np.array([g[4] for g in geoms]), d["F0PI"], *opts,
                                *TTHG, 4, xs)
    for s, g in enumerate(geoms):
        x1 = device.DeviceArray((ng, nt, nwno), ctx)
        resident.reflected_SH(ctx, nlayer + 1, nwno, ng, nt, d, d["surf_reflect"], g[2], g[3], g[4], d["F0PI"], *opts,
                              *TTHG, 4, x1)
        assert np.array_equal(xs[s].to_host(), x1.to_host()), s


def test_batch_randomised_shapes_and_options():
    """Seeded random draws of batch size, grid size (ragged last workgroups), layer count, option set, angle count and
    geometry sharing: every member of every batch equals its single call, reflected and thermal
    (PICASO_FUZZ_OFFSET shifts the seed, as in tests/test_fuzz_gpu.py)."""
    from picaso_amd import _lib, device, resident
    from picaso_amd import synthetic as syn
    ctx = _lib.context()
    rng = np.random.default_rng(4242 + int(os.environ.get("PICASO_FUZZ_OFFSET", "0")))
    for it in range(14):
        B = int(rng.integers(2, 10))
        nwno = int(rng.choice([1, 63, 64, 65, 255, 257, 1000, 4097, 9000, 20011]))
        nlayer = int(rng.choice([1, 2, 3, 17, 60, 90]))
        ng = int(rng.choice([5, 5, 6, 8])) if rng.random() < 0.6 else 3
        phase = 0.0 if ng >= 5 else float(rng.uniform(0.2, 2.5))
        per_geom = bool(rng.random() < 0.4) and phase != 0.0
        opts = [(3, 0, *TTHG), (3, 0, *TTHG), (0, 0, 0.9, -0.8, 2.0, -0.4, 0.9), (1, 1, *TTHG), (2, 0, 1.0, -1.0, 2.0, -0.3, 0.8)][
            int(rng.integers(0, 5))]
        tc = int(rng.integers(0, 2))
        shared_planes = bool(rng.random() < 0.25)
        ctx_, host, devs = _scenes(1 if shared_planes else B, nlayer, nwno, 500 + 20 * it, cloud=bool(rng.random() < 0.7))
        if shared_planes:
            host, devs = host * B, devs * B
        geoms = [_geom(ng, phase + (0.1 * s if per_geom else 0.0)) for s in range(B)]
        if not per_geom:
            geoms = [geoms[0]] * B
        tag = (it, B, nwno, nlayer, ng, phase, per_geom, opts[:2], tc, shared_planes)
        ngg, nt = geoms[0][0], geoms[0][1]
        xs = [device.DeviceArray((ngg, nt, nwno), ctx) for _ in range(B)]
        als = [device.DeviceArray((nwno,), ctx) for _ in range(B)]
        same = all(g is geoms[0] for g in geoms)
        resident.reflected_1d_batch(ctx, nlayer + 1, nwno, ngg, nt, devs, [d["surf_reflect"] for d in devs],
                                    geoms[0][2] if same else np.stack([g[2] for g in geoms]),
                                    geoms[0][3] if same else np.stack([g[3] for g in geoms]),
                                    geoms[0][4] if same else np.array([g[4] for g in geoms]),
                                    [d["F0PI"] for d in devs], *opts, xs, toon_coefficients=tc, b_top=0.01 * it,
                                    gweight=geoms[0][5], tweight=geoms[0][6], albedo=als)
        for s in range(B):
            g = geoms[s]
            x1, a1 = device.DeviceArray((ngg, nt, nwno), ctx), device.DeviceArray((nwno,), ctx)
            resident.reflected_1d(ctx, nlayer + 1, nwno, ngg, nt, devs[s], devs[s]["surf_reflect"], g[2], g[3], g[4],
                                  devs[s]["F0PI"], *opts, x1, toon_coefficients=tc, b_top=0.01 * it, gweight=g[5],
                                  tweight=g[6], albedo=a1)
            assert np.array_equal(xs[s].to_host(), x1.to_host(), equal_nan=True), tag + (s,)
            assert np.array_equal(als[s].to_host(), a1.to_host(), equal_nan=True), tag + (s,)
        # thermal emission on the same scenes
        wno_d = device.DeviceArray.from_host(host[0]["wno"], ctx)
        hard = int(rng.integers(0, 2))
        fs = [device.DeviceArray((ngg, nt, nwno), ctx) for _ in range(B)]
        ds = [device.DeviceArray((nwno,), ctx) for _ in range(B)]
        tl = np.stack([sc["tlevel"] * (1.0 + 0.02 * s) for s, sc in enumerate(host)])
        resident.thermal_1d_batch(ctx, nlayer + 1, wno_d, nwno, ngg, nt, tl, [d["dtau_og"] for d in devs],
                                  [d["w0_no_raman"] for d in devs], [d["cosb_og"] for d in devs],
                                  np.stack([sc["plevel"] for sc in host]),
                                  geoms[0][3] if same else np.stack([g[3] for g in geoms]),
                                  [d["surf_reflect"] for d in devs], hard, fs, gweight=geoms[0][5], tweight=geoms[0][6],
                                  flux_disk=ds)
        for s in range(B):
            g = geoms[s]
            f1, d1 = device.DeviceArray((ngg, nt, nwno), ctx), device.DeviceArray((nwno,), ctx)
            resident.thermal_1d(ctx, nlayer + 1, wno_d, nwno, ngg, nt, tl[s], devs[s]["dtau_og"], devs[s]["w0_no_raman"],
                                devs[s]["cosb_og"], host[s]["plevel"], g[3], devs[s]["surf_reflect"], hard, f1,
                                gweight=g[5], tweight=g[6], flux_disk=d1)
            assert np.array_equal(fs[s].to_host(), f1.to_host(), equal_nan=True), tag + (s, "thermal")
            assert np.array_equal(ds[s].to_host(), d1.to_host(), equal_nan=True), tag + (s, "thermal")
