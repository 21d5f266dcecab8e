"""The climate solver's RT call, get_fluxes (level fluxes + dwni-weighted sums):
oracle/climate_oracle.py and picaso_amd.climate.get_fluxes against tests/golden/climate_fluxes.npz,
outputs of the reference's own climate.get_fluxes with its namedtuples."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, rel_err

CASES = ("a", "holes", "g1")
OUT = ("flux_net_v_layer", "flux_net_v", "flux_plus_v", "flux_minus_v", "flux_net_ir_layer", "flux_net_ir",
       "flux_plus_ir", "flux_minus_ir")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "climate_fluxes.npz"))


def _args(g, c, mod):
    """The reference's positional arguments, built with `mod`'s namedtuples."""
    def tup(prefix):
        p = {k: g["%s/%s%s" % (c, prefix, k)] for k in ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray", "gcos2",
                                                       "w0_no_raman", "dtau_og", "tau_og", "w0_og", "cosb_og")}
        return (mod.OpacityWEd_Tuple(p["dtau"], p["tau"], p["w0"], p["cosb"], p["ftau_cld"], p["ftau_ray"],
                                     p["gcos2"], p["w0_no_raman"], None),
                mod.OpacityNoEd_Tuple(p["dtau_og"], p["tau_og"], p["w0_og"], p["cosb_og"]))
    nlevel, nwno, ngauss = g[c + "/tau"].shape
    atm = mod.Atmosphere_Tuple(None, None, nlevel, g[c + "/tlevel"], g[c + "/plevel"], None, None, None, None)
    wed, noed = tup("")
    sp = mod.ScatteringPhase_Tuple(np.full(nwno, 0.1), 3, 0, 1.0, -1.0, 2.0, -0.5, 1.0)
    dis = mod.Disco_Tuple(5, 1, g[c + "/gweight"], g[c + "/tweight"], g[c + "/ubar0"], g[c + "/ubar1"], 1.0)
    og = mod.Opagrid_Tuple(nwno, g[c + "/dwni"], g[c + "/wno"], ngauss, g[c + "/gauss_wts"])
    kw = {}
    if c == "holes":
        hw, hn = tup("clear/")
        kw = dict(do_holes=True, fhole=0.3, hole_OpacityWEd=hw, hole_OpacityNoEd=hn)
    return (atm, wed, noed, sp, dis, og, g[c + "/f0pi"], True, True), kw


def _check(out, g, c, tol, tol_ir_lvl=None):
    """Net fluxes: relative to the field maximum.  Level fluxes per wavenumber: tests/helpers.lvl_err's
    metric, |got - ref| over the per-wavenumber scale of the (plus, minus) pair -- entries that all
    but vanish are differences of nearly equal terms in the reference's formulation.  The thermal
    level fluxes of optically thick Gauss points carry the reference's own b_surface - c_plus_down
    cancellation (tests/test_ck_gpu.py): `tol_ir_lvl`."""
    assert len(out) == 8
    got = dict(zip(OUT, out))
    for name in OUT:
        want = g["%s/out/%s" % (c, name)]
        assert np.shape(got[name]) == want.shape, name
    for name in ("flux_net_v_layer", "flux_net_v", "flux_net_ir_layer", "flux_net_ir"):
        want = g["%s/out/%s" % (c, name)]
        assert rel_err(got[name], want, 1e-4 * np.abs(want).max()) < tol, (c, name)
    for leg, t_ in (("v", tol), ("ir", tol_ir_lvl or tol)):
        wp, wm = g["%s/out/flux_plus_%s" % (c, leg)], g["%s/out/flux_minus_%s" % (c, leg)]
        ax = tuple(range(wp.ndim - 1))
        scale = np.maximum(np.abs(wp).max(axis=ax), np.abs(wm).max(axis=ax))
        scale = np.where(scale == 0, 1.0, scale)
        for nm, w in (("plus", wp), ("minus", wm)):
            assert np.max(np.abs(got["flux_%s_%s" % (nm, leg)] - w) / scale) < t_, (c, leg, nm)


@pytest.mark.parametrize("c", CASES)
def test_oracle_get_fluxes(gold, c):
    from oracle import climate_oracle as co
    from picaso_amd import climate as pc          # namedtuple definitions only (no device call)
    args, kw = _args(gold, c, pc)
    _check(co.get_fluxes(*args, **kw), gold, c, 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES)
def test_gpu_get_fluxes(gold, c):
    from picaso_amd import climate as pc
    args, kw = _args(gold, c, pc)
    _check(pc.get_fluxes(*args, **kw), gold, c, 2e-8, tol_ir_lvl=2e-4)


@pytest.mark.gpu
def test_gpu_get_fluxes_single_legs(gold):
    """reflected-only / thermal-only calls leave the other leg's outputs at zero, as the reference."""
    from picaso_amd import climate as pc
    args, kw = _args(gold, "a", pc)
    full = pc.get_fluxes(*args, **kw)
    r = pc.get_fluxes(*args[:7], True, False)
    t = pc.get_fluxes(*args[:7], False, True)
    # the (ng, nt, ...) visible fluxes are broadcast views of the single two-stream result by default
    assert not full[2].flags.writeable and full[2].strides[:2] == (0, 0)
    cp = pc.get_fluxes(*args, copy_outputs=True, **kw)
    assert cp[2].flags.writeable and np.array_equal(cp[2], full[2]) and np.array_equal(cp[3], full[3])
    for k in range(4):
        assert np.array_equal(r[k], full[k]) and not np.any(t[k])
        assert np.array_equal(t[4 + k], full[4 + k]) and not np.any(r[4 + k])


@pytest.mark.gpu
def test_gpu_calculate_atm_feeds_get_fluxes(oracle):
    """climate.calculate_atm (resident planes) -> climate.get_fluxes, on a 4-point correlated-k table:
    planes against the reference's compute_opacity(ngauss=4) fixture, fluxes against the oracle's
    get_fluxes on those reference planes."""
    import test_ck_optics as tck
    from oracle import climate_oracle as co
    from picaso_amd import climate as pc
    from picaso_amd import justdoit as jdi
    ck = np.load(os.path.join(GOLDEN, "ck.npz"))
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    opa = tck._ck_class(ck)
    case = tck._case(og, jdi, True)
    wed, noed, sp, dis, atm_t, holes = pc.calculate_atm(case, opa)
    assert holes == (None, None) and sp.surf_reflect == 0 and atm_t.nlevel == len(og["in/tlevel"])
    ref = {nm: ck["de1_s2/" + nm] for nm in tck.NAMES}
    for got, nm in ((wed.DTAU, "dtau"), (wed.TAU, "tau"), (wed.W0, "w0"), (wed.COSB, "cosb"), (wed.GCOS2, "gcos2"),
                    (wed.W0_no_raman, "w0_no_raman"), (noed.DTAU, "dtau_og"), (noed.TAU, "tau_og")):
        assert tck._close(got.to_host(), ref[nm], 1e-9), nm
    assert np.allclose(atm_t.dtdp, np.diff(np.log(og["in/tlevel"])) / np.diff(np.log(og["in/plevel_bar"])))
    only = pc.calculate_atm(case, opa, only_atmosphere=True)
    assert np.array_equal(only.t_level, atm_t.t_level) and only.condensables == ["H2O", "CH4"]
    nwno, ngauss = opa.nwno, opa.ngauss
    grid = pc.Opagrid_Tuple(nwno, np.abs(np.gradient(opa.wno)), opa.wno, ngauss, ck["in/gauss_wts"])
    out = pc.get_fluxes(atm_t, wed, noed, sp, dis, grid, np.ones(nwno), True, True)
    wed_h = pc.OpacityWEd_Tuple(ref["dtau"], ref["tau"], ref["w0"], ref["cosb"], ref["ftau_cld"], ref["ftau_ray"],
                                ref["gcos2"], ref["w0_no_raman"], None)
    noed_h = pc.OpacityNoEd_Tuple(ref["dtau_og"], ref["tau_og"], ref["w0_og"], ref["cosb_og"])
    want = co.get_fluxes(atm_t, wed_h, noed_h, sp, dis, grid, np.ones(nwno), True, True)
    # visible nets tight; the thermal nets of this scene's optically thick Gauss points carry the
    # reference formulation's own cancellation noise at the deep levels (tests/test_ck_gpu.py)
    for name, a, b in zip(OUT, out, want):
        if name.startswith("flux_net"):
            assert rel_err(a, b, 1e-4 * np.abs(b).max()) < (1e-4 if name.endswith("_ir") or "ir_" in name else 1e-7), name


@pytest.mark.gpu
@pytest.mark.parametrize("c", ["a", "g1"])
def test_gpu_get_fluxes_tbatch_is_the_jacobian_loop(gold, c):
    """climate.get_fluxes_tbatch: the thermal leg for many level-temperature profiles over ONE set of opacities -- what
    the reference's Jacobian asks of get_fluxes one perturbed profile at a time (climate.py:1105-1180).  Profiles as
    the solver makes them (one level raised by max(1e-4 T, 3 K), the rest unchanged) plus two globally different
    ones; every row equals get_fluxes on that profile bit for bit, in one chunk and in ragged chunks."""
    from picaso_amd import climate as clim
    (atm, wed, noed, sp, dis, og, f0, _, _), kw = _args(gold, c, clim)
    t0 = np.asarray(atm.t_level, dtype=float)
    nlevel = len(t0)
    profiles = [t0]
    for jm in range(0, nlevel, max(1, nlevel // 7)):
        t = t0.copy()
        t[jm] += max(1e-4 * t0[jm], 3.0)
        profiles.append(t)
    profiles += [t0 * 1.05, t0[::-1].copy()]
    temps = np.stack(profiles)
    ctx = clim._lib.context()
    # resident planes, as calculate_atm hands them over
    wed_d = clim.OpacityWEd_Tuple(*[clim.DeviceArray.from_host(x, ctx) if x is not None else None for x in wed])
    noed_d = clim.OpacityNoEd_Tuple(*[clim.DeviceArray.from_host(x, ctx) for x in noed])
    want = [clim.get_fluxes(atm._replace(t_level=t), wed_d, noed_d, sp, dis, og, f0, False, True, ctx=ctx)[4:] for t in temps]
    for chunk in (64, 3):
        got = clim.get_fluxes_tbatch(temps, atm, wed_d, noed_d, sp, dis, og, ctx=ctx, chunk=chunk)
        for k in range(len(temps)):
            for j, name in enumerate(("flux_net_ir_layer", "flux_net_ir", "flux_plus_ir", "flux_minus_ir")):
                assert np.array_equal(got[j][k], want[k][j]), (c, chunk, k, name)
    nl, nn = clim.get_fluxes_tbatch(temps, atm, wed_d, noed_d, sp, dis, og, ctx=ctx, chunk=5, nets_only=True)
    for k in range(len(temps)):                       # sums on the device: a fixed tree instead of numpy's order
        for got_, j in ((nl, 0), (nn, 1)):
            w_ = want[k][j]
            assert np.max(np.abs(got_[k] - w_)) <= 1e-13 * np.abs(w_).max(), (c, k, j)
    with pytest.raises(Exception, match="nlevel"):
        clim.get_fluxes_tbatch(temps[:, :-1], atm, wed_d, noed_d, sp, dis, og, ctx=ctx)
