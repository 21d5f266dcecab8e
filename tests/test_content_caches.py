"""The content caches key device copies by a digest of EVERY byte of the host array (optics.content_digest): the
reference re-reads its inputs on every call (justdoit.py:437-449, atmsetup.py:609-622, deq_chem.py:334-384), so an
in-place edit of one (layer, facet) row or of one element must be seen every time -- round 4's strided samples missed
29-100 % of such edits."""
import os

import numpy as np
import pytest

from helpers import GOLDEN


def test_cloud_table_fingerprint_sees_every_row_rewrite():
    from picaso_amd.optics import _table_fingerprint
    rng = np.random.default_rng(3)
    nlayer, nin, nfac = 90, 196, 64
    tabs = [rng.random((nlayer, nin, nfac)) for _ in range(3)]
    wn = np.linspace(100.0, 30000.0, nin)
    stamp = _table_fingerprint(tabs, wn)
    seen = 0
    for trial in range(270):
        lay, fac = rng.integers(nlayer), rng.integers(nfac)
        tabs[trial % 3][lay, :, fac] = rng.random(nin)
        new = _table_fingerprint(tabs, wn)
        seen += new != stamp
        stamp = new
    assert seen == 270


def test_cloud_table_fingerprint_sees_every_single_element_edit():
    from picaso_amd.optics import _table_fingerprint
    rng = np.random.default_rng(4)
    tabs = [rng.random((90, 196)) for _ in range(3)]
    wn = np.linspace(100.0, 30000.0, 196)
    stamp = _table_fingerprint(tabs, wn)
    seen = 0
    for trial in range(90):
        t = tabs[trial % 3]
        i, j = rng.integers(90), rng.integers(196)
        t[i, j] = np.nextafter(t[i, j], 2.0)              # the smallest edit there is
        new = _table_fingerprint(tabs, wn)
        seen += new != stamp
        stamp = new
    assert seen == 90
    wn[17] = np.nextafter(wn[17], 0.0)                    # ... and the grid counts too
    assert _table_fingerprint(tabs, wn) != stamp


def test_ktable_fingerprint_sees_every_single_element_edit():
    from picaso_amd.deq_chem import _fingerprint
    rng = np.random.default_rng(5)
    k = -50.0 + 10.0 * rng.random((20, 30, 661, 8))
    stamp = _fingerprint(k)
    seen = 0
    for _ in range(200):
        idx = tuple(rng.integers(n) for n in k.shape)
        k[idx] = np.nextafter(k[idx], 0.0)
        new = _fingerprint(k)
        seen += new != stamp
        stamp = new
    assert seen == 200
    # same content at another address, or seen through a non-contiguous view: the same table
    assert _fingerprint(k.copy()) == stamp
    assert _fingerprint(np.asfortranarray(k)) == stamp
    assert _fingerprint(k.reshape(600, 661, 8)) != stamp          # the shape is part of the identity


def test_digest_fallback_without_xxhash(monkeypatch):
    from picaso_amd import optics
    a = np.arange(1000.0)
    d1 = optics.content_digest(a)
    monkeypatch.setattr(optics, "_xxh3", None)
    d2 = optics.content_digest(a)
    a[500] += 1e-9
    assert len(d1) == len(d2) == 16 and optics.content_digest(a) != d2


@pytest.mark.gpu
def test_gpu_mix_all_gases_sees_in_place_edit_of_one_coefficient():
    """deq_chem.mix_all_gases_gasesfly keeps the per-gas tables in HBM between calls: one coefficient changed in place
    (same address, same shape) must give the result of a fresh array holding the edited values."""
    from picaso_amd import deq_chem
    g = np.load(os.path.join(GOLDEN, "mixing.npz"))
    c = "g8"
    kappas = [np.array(k) for k in g[c + "/kappas"]]
    args = ([m for m in g[c + "/mixes"]], g[c + "/gauss_pts"], g[c + "/gauss_wts"], g[c + "/indices"])
    first = deq_chem.mix_all_gases_gasesfly(kappas, *args)
    assert np.max(np.abs(first - g[c + "/kappa_mixed"])) < 1e-11
    idx = g[c + "/indices"]
    rng = np.random.default_rng(1)
    for trial in range(20):
        gas = int(rng.integers(len(kappas)))
        lay = int(rng.integers(idx.shape[-1] if idx.ndim == 2 else len(idx[0])))
        ip, it = int(np.asarray(idx[0]).ravel()[lay]), int(np.asarray(idx[2]).ravel()[lay])
        w, q = int(rng.integers(kappas[gas].shape[2])), int(rng.integers(kappas[gas].shape[3]))
        for k in (kappas if trial % 2 else [kappas[gas]]):         # ln kappa: a factor 1.65 on one coefficient
            k[ip, it, w, q] += 0.5
        second = deq_chem.mix_all_gases_gasesfly(kappas, *args)
        deq_chem.clear_table_cache()
        fresh = deq_chem.mix_all_gases_gasesfly([k.copy() for k in kappas], *args)
        assert np.array_equal(second, fresh), trial
        if trial % 2:                                              # every gas raised at a point the layer reads: visible
            assert not np.array_equal(second, first), trial
        first = second
