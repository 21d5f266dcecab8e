"""get_thermal_SH (reference fluxes.py:2979-3182) with the angle-independent block algebra shared between the disk
angles of a lane (`k_sh_thermal<NB, NA>`, sh.hip): against the oracle on fresh scenes, against the round-2 kernel
(one wave per angle, `PICASO_AMD_SH_THERMAL_PER_ANGLE=1`), and the same bits however many angles share a lane
(`PICASO_AMD_SHT_ANGLES=1..5`) and however the wavelengths are cut into blocks."""
import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from picaso_amd import _lib, disco, fluxes
    assert _lib.device_count() > 0, "no MI355X visible"
    _lib.context()

    class H:
        pass
    h = H()
    h.fluxes, h.disco = fluxes, disco
    return h


def _scene(nlayer, nwno, seed, stream, cloud=True):
    from picaso_amd import synthetic as syn
    return syn.make_scene(nlayer, nwno, seed=seed, stream=stream, cloud=cloud)


def _geom(hip, ng, nt, phase=0.0):
    if nt == 1:
        g, gw, t, tw = hip.disco.get_angles_1d(ng)
    else:
        g, gw, t, tw = hip.disco.get_angles_3d(ng, nt)
    u0, u1, ct, _, _ = hip.disco.compute_disco(ng, nt, g, t, phase)
    return u1


def _args(sc, nlayer, nwno, ng, nt, u1, rs, stream, hard, delta=True):
    cosb = sc["cosb"] if delta else sc["cosb_og"]
    return (nlayer + 1, sc["wno"], nwno, ng, nt, sc["tlevel"], sc["dtau"], sc["tau"], sc["w0"], cosb, sc["dtau_og"],
            sc["tau_og"], sc["w0_og"], sc["w0_no_raman"], sc["cosb_og"], sc["plevel"], u1, rs, stream, hard)


@pytest.mark.parametrize("stream", [2, 4])
@pytest.mark.parametrize("ng,nt", [(5, 1), (6, 1), (8, 1), (3, 2), (4, 3)])
def test_against_oracle_and_per_angle_kernel(hip, oracle, monkeypatch, stream, ng, nt):
    nlayer, nwno = 33, 517
    rng = np.random.default_rng(100 * ng + 10 * nt + stream)
    monkeypatch.delenv("PICASO_AMD_SH_THERMAL_PER_ANGLE", raising=False)
    monkeypatch.delenv("PICASO_AMD_SHT_ANGLES", raising=False)
    for trial, (cloud, hard, delta) in enumerate([(True, 0, True), (False, 1, True), (True, 1, False)]):
        sc = _scene(nlayer, nwno, 400 + 7 * trial + ng, stream, cloud=cloud)
        u1 = _geom(hip, ng, nt, 0.0 if nt == 1 else 0.7)
        rs = 0.3 * rng.random(nwno) if trial else 0.0
        args = _args(sc, nlayer, nwno, ng, nt, u1, rs, stream, hard, delta)
        got, _ = hip.fluxes.get_thermal_SH(*args)
        want, _ = oracle.get_thermal_SH(*args)
        assert rel_err(got, want) < 1e-9, (trial, "oracle")
        monkeypatch.setenv("PICASO_AMD_SH_THERMAL_PER_ANGLE", "1")
        old, _ = hip.fluxes.get_thermal_SH(*args)
        monkeypatch.delenv("PICASO_AMD_SH_THERMAL_PER_ANGLE")
        assert rel_err(got, old) < 1e-11, (trial, "per-angle kernel")
        for m in (1, 2, 3, 4, 5):                       # angles per lane: same bits
            monkeypatch.setenv("PICASO_AMD_SHT_ANGLES", str(m))
            alt, _ = hip.fluxes.get_thermal_SH(*args)
            assert np.array_equal(alt, got), (trial, m)
        monkeypatch.delenv("PICASO_AMD_SHT_ANGLES")


@pytest.mark.parametrize("stream", [2, 4])
def test_thick_thin_and_single_layer(hip, oracle, stream):
    """One and two layers, optically thin and thick (35-clipped) columns, conservative scattering."""
    ng, nt = 5, 1
    u1 = _geom(hip, ng, nt)
    for nlayer, scale in ((1, 1.0), (2, 1e-4), (7, 300.0), (12, 1.0)):
        nwno = 130
        sc = dict(_scene(nlayer, nwno, 900 + nlayer, stream))
        for k in ("dtau", "dtau_og"):
            sc[k] = sc[k] * scale
        for k in ("tau", "tau_og"):
            sc[k] = sc[k] * scale
        if nlayer == 12:
            sc["w0"] = np.minimum(sc["w0"] * 0 + 0.999999, 0.999999)
        args = _args(sc, nlayer, nwno, ng, nt, u1, 0.1, stream, 0)
        got, _ = hip.fluxes.get_thermal_SH(*args)
        want, _ = oracle.get_thermal_SH(*args)
        assert np.isfinite(got).all()
        assert rel_err(got, want) < 1e-8, (nlayer, scale)


def test_full_size_blocks_are_the_whole(hip, oracle):
    """1e5 wavelengths x 90 layers x 5 angles (five angles per lane) in 8 blocks of 12 500 (one angle per lane):
    np.array_equal, and sampled columns against the oracle."""
    from picaso_amd import synthetic as syn
    nlayer, nwno, ng, nt = 90, 100000, 5, 1
    sc = syn.make_scene(nlayer, nwno, seed=3, stream=4)
    u1 = _geom(hip, ng, nt)
    whole, _ = hip.fluxes.get_thermal_SH(*_args(sc, nlayer, nwno, ng, nt, u1, 0.0, 4, 0))
    cut = np.linspace(0, nwno, 9).astype(int)
    for lo, hi in zip(cut[:-1], cut[1:]):
        sub = {k: (np.ascontiguousarray(v[..., lo:hi]) if np.ndim(v) and np.shape(v)[-1] == nwno else v)
               for k, v in sc.items()}
        part, _ = hip.fluxes.get_thermal_SH(*_args(sub, nlayer, hi - lo, ng, nt, u1, 0.0, 4, 0))
        assert np.array_equal(part, whole[..., lo:hi]), (lo, hi)
    idx = np.arange(0, nwno, 997)
    sub = {k: (np.ascontiguousarray(v[..., idx]) if np.ndim(v) and np.shape(v)[-1] == nwno else v) for k, v in sc.items()}
    want, _ = oracle.get_thermal_SH(*_args(sub, nlayer, idx.size, ng, nt, u1, 0.0, 4, 0))
    assert rel_err(whole[..., idx], want) < 1e-9
