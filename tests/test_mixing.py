"""On-the-fly correlated-k gas mixing (resort-rebin; SURVEY.md 8f rank 1 follow-up):
oracle/mix_oracle.c and the HIP kernel against tests/golden/mixing.npz, outputs of the reference's
own deq_chem.mix_all_gases_gasesfly / mix_2_gases (tests/golden/make_golden.py mixing)."""
import os

import numpy as np
import pytest

from helpers import GOLDEN

CASES = ("g8", "g4", "two", "g8many")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "mixing.npz"))


def _case(g, c):
    return ([k for k in g[c + "/kappas"]], [m for m in g[c + "/mixes"]], g[c + "/gauss_pts"],
            g[c + "/gauss_wts"], g[c + "/indices"])


@pytest.mark.parametrize("c", CASES)
def test_oracle_mixing(gold, oracle, c):
    out = oracle.mix_all_gases_gasesfly(*_case(gold, c))
    want = gold[c + "/kappa_mixed"]
    assert out.shape == want.shape
    # ln(kappa): absolute error = relative error of the mixed k-coefficient
    assert np.max(np.abs(out - want)) < 1e-12


def _gpu_mix(g, c):
    from picaso_amd import _lib, resident
    from picaso_amd.device import DeviceArray
    ctx = _lib.context()
    kappas, mixes, pts, wts, idx = _case(g, c)
    dk = [DeviceArray.from_host(k, ctx) for k in kappas]
    out = resident.mix_all_gases_gasesfly(ctx, dk, mixes, pts, wts, idx).to_host()
    return np.moveaxis(out, 1, 3)            # (nlayer, 4, nwno, nk) -> the reference's (nlayer, nwno, nk, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES)
def test_gpu_mixing_vs_reference(gold, c):
    out = _gpu_mix(gold, c)
    want = gold[c + "/kappa_mixed"]
    assert out.shape == want.shape
    assert np.max(np.abs(out - want)) < 1e-11       # ln(kappa): relative error of the mixed coefficient


@pytest.mark.gpu
def test_gpu_mixing_vs_oracle_large(oracle):
    """Climate-table shape: 8 Gauss points, 12 gases, 90 layers x 211 bins (ragged last block)."""
    rng = np.random.default_rng(7)
    nk, ngas, npres, ntemp, nwno, nlayer = 8, 12, 6, 5, 211, 90
    xg, wg = np.polynomial.legendre.leggauss(4)
    pts = np.concatenate([0.95 * 0.5 * (xg + 1), 0.95 + 0.05 * 0.5 * (xg + 1)])
    wts = np.concatenate([0.95 * 0.5 * wg, 0.05 * 0.5 * wg])
    kappas = [-55.0 + 10.0 * rng.random((npres, ntemp, nwno, 1))
              + np.cumsum(rng.random((npres, ntemp, nwno, nk)) * rng.choice([0.1, 1.0, 5.0]), axis=3)
              for _ in range(ngas)]
    mixes = [10.0 ** (-1.0 - 7.0 * rng.random(nlayer)) for _ in range(ngas)]
    p_low, t_low = rng.integers(0, npres - 1, nlayer), rng.integers(0, ntemp - 1, nlayer)
    idx = np.array([p_low, p_low + 1, t_low, t_low + 1])
    want = oracle.mix_all_gases_gasesfly(kappas, mixes, pts, wts, idx)
    g = {"x/kappas": np.stack(kappas), "x/mixes": np.stack(mixes), "x/gauss_pts": pts, "x/gauss_wts": wts,
         "x/indices": idx}
    out = _gpu_mix(g, "x")
    assert np.max(np.abs(out - want)) < 1e-11


@pytest.mark.gpu
def test_gpu_mixing_rejects_bad_arguments():
    from picaso_amd import _lib, resident
    from picaso_amd.device import DeviceArray
    ctx = _lib.context()
    k = [DeviceArray.from_host(np.zeros((2, 2, 3, 9)), ctx) for _ in range(2)]
    idx = np.array([[0], [1], [0], [1]])
    with pytest.raises(Exception, match="ngauss"):
        resident.mix_all_gases_gasesfly(ctx, k, [np.ones(1)] * 2, np.linspace(0.1, 0.9, 9), np.ones(9) / 9, idx)
    k = [DeviceArray.from_host(np.zeros((2, 2, 3, 4)), ctx) for _ in range(2)]
    with pytest.raises(Exception, match="outside"):
        resident.mix_all_gases_gasesfly(ctx, k, [np.ones(1)] * 2, np.linspace(0.1, 0.9, 4), np.ones(4) / 4,
                                        np.array([[0], [2], [0], [1]]))


@pytest.mark.gpu
def test_gpu_mixing_reference_signature(gold):
    """picaso_amd.deq_chem.mix_all_gases_gasesfly: numpy in, the reference's array layout out; the
    device copies of the tables are reused until their content changes."""
    from picaso_amd import deq_chem
    deq_chem.clear_table_cache()
    args = _case(gold, "g8")
    out = deq_chem.mix_all_gases_gasesfly(*args)
    want = gold["g8/kappa_mixed"]
    assert out.shape == want.shape and np.max(np.abs(out - want)) < 1e-11
    n_cached = len(deq_chem._tables)
    assert n_cached == len(args[0])
    out2 = deq_chem.mix_all_gases_gasesfly(*args)
    assert np.array_equal(out, out2) and len(deq_chem._tables) == n_cached
    args[0][1][...] = args[0][1] + 0.5               # same buffer, new content: re-uploaded
    out3 = deq_chem.mix_all_gases_gasesfly(*args)
    assert not np.array_equal(out, out3)


@pytest.mark.gpu
def test_gpu_mixing_fuzz_vs_oracle(oracle):
    """Random Gauss-point counts 1..8 (odd squares leave a ragged last key pair), 1..10 gases, flat
    gases (8-way ties in every sort), tiny and huge abundances, sizes that leave partial blocks."""
    rng = np.random.default_rng(31337 + int(os.environ.get("PICASO_FUZZ_OFFSET", "0")))
    for it in range(30):
        nk = int(rng.integers(1, 9))
        ngas = int(rng.integers(1, 11))
        npres, ntemp = int(rng.integers(2, 5)), int(rng.integers(2, 5))
        nwno, nlayer = int(rng.choice([1, 3, 4, 5, 33])), int(rng.integers(1, 6))
        xg, wg = np.polynomial.legendre.leggauss(nk)
        pts, wts = 0.5 * (xg + 1), 0.5 * wg
        kappas = []
        for g in range(ngas):
            base = -60.0 + 20.0 * rng.random((npres, ntemp, nwno, 1))
            kind = rng.integers(0, 3)
            if kind == 0:                                   # flat: every coefficient of a bin equal
                k = np.repeat(base, nk, axis=3)
            elif kind == 1:                                 # steep
                k = base + np.cumsum(rng.random((npres, ntemp, nwno, nk)) * 8.0, axis=3)
            else:                                           # gentle, with repeated values
                k = base + np.cumsum(np.round(rng.random((npres, ntemp, nwno, nk)) * 2.0) * 0.25, axis=3)
            kappas.append(np.ascontiguousarray(k))
        mixes = [10.0 ** rng.uniform(-12, 0, nlayer) for _ in range(ngas)]
        p_low, t_low = rng.integers(0, npres - 1, nlayer), rng.integers(0, ntemp - 1, nlayer)
        idx = np.array([p_low, p_low + 1, t_low, t_low + 1])
        want = oracle.mix_all_gases_gasesfly(kappas, mixes, pts, wts, idx)
        g = {"x/kappas": np.stack(kappas), "x/mixes": np.stack(mixes), "x/gauss_pts": pts, "x/gauss_wts": wts,
             "x/indices": idx}
        out = _gpu_mix(g, "x")
        assert out.shape == want.shape
        assert np.max(np.abs(out - want)) < 1e-10, (it, nk, ngas, nwno, nlayer)
