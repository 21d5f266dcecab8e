"""Transmission spectrum get_transit_1d: CPU oracle and HIP kernel against
tests/golden/transit.npz (outputs of the reference's own fluxes.get_transit_1d)."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, rel_err

CASES = ("a", "b", "two")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "transit.npz"))


def _args(g, c):
    nlevel, nwno = g[c + "/z"].size, g[c + "/dtau"].shape[1]
    return (g[c + "/z"], g[c + "/dz"], nlevel, nwno, float(g[c + "/rstar"]), g[c + "/mmw"],
            float(g[c + "/k_b"]), float(g[c + "/amu"]), g[c + "/plevel"], g[c + "/tlevel"],
            g[c + "/colden"], g[c + "/dtau"])


@pytest.mark.parametrize("c", CASES)
def test_oracle_transit(gold, oracle, c):
    # the transit depth is (zmin/Rs)^2 + a small atmospheric term: compare the atmospheric part too
    F = oracle.get_transit_1d(*_args(gold, c))
    base = (gold[c + "/z"].min() / float(gold[c + "/rstar"])) ** 2
    assert rel_err(F, gold[c + "/F"]) < 1e-13
    assert rel_err(F - base, gold[c + "/F"] - base) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES)
def test_gpu_transit(gold, c):
    from picaso_amd import fluxes
    F = fluxes.get_transit_1d(*_args(gold, c))
    base = (gold[c + "/z"].min() / float(gold[c + "/rstar"])) ** 2
    assert rel_err(F, gold[c + "/F"]) < 1e-13
    assert rel_err(F - base, gold[c + "/F"] - base) < 1e-9


@pytest.mark.gpu
def test_gpu_transit_vs_oracle_large(oracle):
    """90 layers x 5003 wavelengths (ragged last block), resident entry point."""
    import ctypes
    from picaso_amd import _lib, synthetic as syn
    from picaso_amd._lib import check, f64, load, ptr
    from picaso_amd.device import DeviceArray
    nlayer, nwno = 90, 5003
    sc = syn.make_scene(nlayer, nwno, seed=19)
    k_b, amu = 1.380649e-16, 1.66053906660e-24
    p, t = sc["plevel"], sc["tlevel"]
    mmw = np.full(nlayer, 2.3)
    H = k_b * 0.5 * (t[1:] + t[:-1]) / (mmw * amu * 2500.0)
    dzl = H * np.log(p[1:] / p[:-1])
    z = 7.0e9 + np.concatenate([np.cumsum(dzl[::-1])[::-1], [0.0]])
    dz = np.concatenate([dzl, [dzl[-1]]])
    colden = (p[1:] - p[:-1]) / 2500.0
    args = (z, dz, nlayer + 1, nwno, 6.96e10, mmw, k_b, amu, p, t, colden, sc["dtau_og"])
    Fo = oracle.get_transit_1d(*args)
    ctx = _lib.context()
    d = DeviceArray.from_host(sc["dtau_og"], ctx)
    out = DeviceArray((nwno,), ctx)
    check(load().picaso_get_transit_1d_dev(
        ctx, ptr(f64(z)), ptr(f64(dz)), ctypes.c_int(nlayer + 1), ctypes.c_int(nwno), ctypes.c_long(nwno),
        ctypes.c_double(6.96e10), ptr(f64(mmw)), ctypes.c_double(k_b), ctypes.c_double(amu), ptr(f64(p)),
        ptr(f64(t)), ptr(f64(colden)), ptr(d.addr), ptr(out.addr)), ctx)
    base = (z.min() / 6.96e10) ** 2
    assert rel_err(out.to_host() - base, Fo - base) < 1e-9
