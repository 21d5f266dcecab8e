"""picaso_host_setup (csrc/setup.hip, picaso_amd/fastsetup.py) against the numpy mirror it replaces -- ATMSETUP
(reference atmsetup.py:74-461), get_opacities (optics.py:2048-2123, 2241-2306) and the per-layer coefficients
(optics.py:144-277): every array bit for bit on random profiles.  No GPU: the function is host code; the opacity object's
device tables are stand-ins (the stages under test never touch them)."""
import os

import numpy as np
import pytest

from picaso_amd import _lib, fastsetup, justdoit as jdi, optics as px

pytestmark = pytest.mark.skipif(not os.path.exists(_lib.LIB_PATH), reason="libpicaso_hip.so not built")


class _Dev:
    addr = 0

    def __init__(self, *a, **k):
        pass

    @classmethod
    def from_host(cls, a, ctx=None):
        return cls()


@pytest.fixture()
def opa(monkeypatch):
    monkeypatch.setattr(px, "DeviceArray", _Dev)
    nwno = 16
    wno = np.linspace(2000.0, 33333.0, nwno)
    temps, press = [100.0, 300.0, 700.0, 1500.0, 3000.0], [1e-6, 1e-4, 1e-2, 1e-1, 1.0, 10.0, 100.0, 500.0]
    pt = [(i + 1, p, t) for i, (t, p) in enumerate((t, p) for t in temps for p in press)]
    molecular = {m: {i: np.ones(nwno) for (i, p, t) in pt} for m in ["H2O", "CH4", "CO", "NH3", "H2"]}
    cia_t = [75.0, 200.0, 500.0, 1000.0, 2000.0, 4000.0]
    continuum = {pr: {t: np.ones(nwno) for t in cia_t} for pr in ("H2H2", "H2He", "H2CH4")}
    ray = {m: np.ones(nwno) for m in ("H2", "He")}
    o = px.RetrieveOpacities(wno, pt, molecular, continuum, cia_t, rayleigh_opa=ray, query_method="linear", ctx=object())
    o._wno_test = wno
    return o


def _case(rng, nlevel, p_reference, cols, planet=False):
    lo, hi = rng.uniform(-7, -4), rng.uniform(0.5, 2.9)
    plev = np.sort(10.0 ** (np.linspace(lo, hi, nlevel) + rng.uniform(-0.01, 0.01, nlevel)))
    prof = {"pressure": plev, "temperature": rng.uniform(60.0, 3500.0) * (0.3 + rng.random(nlevel))}
    mix = rng.random((len(cols), nlevel)) * 10.0 ** rng.uniform(-8, 0, (len(cols), 1))
    for k, v in zip(cols, mix):
        prof[k] = v
    case = jdi.inputs()
    case.phase_angle(0)
    if planet:                                   # radius and mass: gravity G M / z^2 level by level
        case.gravity(radius=float(rng.uniform(3e8, 1.2e10)), mass=float(10.0 ** rng.uniform(27, 30.5)))
    else:
        case.gravity(gravity=float(rng.uniform(300.0, 6000.0)))
    case.atmosphere(df=prof)
    case.approx(raman="none", p_reference=p_reference)
    return case


def _both(case, opa, monkeypatch):
    wno = opa._wno_test
    fast = jdi._setup_atmosphere(case.inputs, opa, wno)
    assert getattr(fast, "_fast", None) is not None, "the C set-up declined a profile inside its scope"
    opa.get_opacities(fast, exclude_mol=1)
    plan_f, fac_f = opa._plan, px._layer_factors(fast, opa)
    monkeypatch.setenv("PICASO_AMD_PY_SETUP", "1")
    ref = jdi._setup_atmosphere(case.inputs, opa, wno)
    assert getattr(ref, "_fast", None) is None
    opa.get_opacities(ref, exclude_mol=1)
    plan_r, fac_r = opa._plan, px._layer_factors(ref, opa)
    monkeypatch.delenv("PICASO_AMD_PY_SETUP")
    return (fast, plan_f, fac_f), (ref, plan_r, fac_r)


def _same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    assert np.array_equal(a, b, equal_nan=True), what


@pytest.mark.parametrize("seed", range(24))
def test_fast_setup_bits_of_the_mirror(opa, monkeypatch, seed):
    rng = np.random.default_rng(100 + seed)
    cols_all = ["H2", "He", "H2O", "CH4", "CO", "NH3", "Na", "K", "TiO", "CO2"]
    cols = list(rng.permutation(cols_all)[:int(rng.integers(3, len(cols_all) + 1))])
    if "H2" not in cols:
        cols[0] = "H2"
    nlevel = int(rng.choice([2, 3, 10, 61, 91]))
    p_ref = float(rng.choice([1.0, 1e-9, 1e5, 10.0 ** rng.uniform(-6, 2)]))
    (f, pf, ff), (r, pr, fr) = _both(_case(rng, nlevel, p_ref, cols, planet=seed % 2 == 1), opa, monkeypatch)
    for d in ("level", "layer"):
        fd, rd = getattr(f, d), getattr(r, d)
        assert set(fd) == set(rd), (d, set(fd) ^ set(rd))
        for k in rd:
            if k == "mixingratios":
                assert list(fd[k]) == list(rd[k])
                for m in rd[k]:
                    _same(fd[k][m], rd[k][m], (d, k, m))
            elif k == "cloud":
                assert set(fd[k]) == set(rd[k])
            else:
                _same(fd[k], rd[k], (d, k))
    assert list(f.molecules) == list(r.molecules) and f.molecules.dtype == r.molecules.dtype
    assert f.continuum_molecules == r.continuum_molecules and f.rayleigh_molecules == r.rayleigh_molecules
    assert f.weights == r.weights and f.warnings == r.warnings
    assert (f.c.nlevel, f.c.nlayer, f.cloud_free, f.hard_surface, f.get_lvl_flux) == \
           (r.c.nlevel, r.c.nlayer, r.cloud_free, r.hard_surface, r.get_lvl_flux)
    assert pf["molecules"] == pr["molecules"] and pf["cia_pairs"] == pr["cia_pairs"] and pf["nlayer"] == pr["nlayer"]
    for k in ("rows", "wts", "fac", "cia_rows"):
        _same(pf[k], pr[k], ("plan", k))
    assert ff[2] == fr[2]
    for i in (0, 1, 3):
        _same(ff[i], fr[i], ("factors", i))


def test_fast_setup_declines_what_it_does_not_cover(opa, monkeypatch):
    rng = np.random.default_rng(5)
    wno = opa._wno_test
    case = _case(rng, 31, 1.0, ["H2", "He", "H2O"])
    assert fastsetup.setup(case.inputs, opa, wno) is not None
    c3 = _case(rng, 31, 1.0, ["H2", "He", "H2O", "e-"])
    assert fastsetup.setup(c3.inputs, opa, wno) is None
    c4 = _case(rng, 31, 1.0, ["H2", "He", "H2O"])
    c4.inputs["atmosphere"]["profile"]["pressure"] = c4.inputs["atmosphere"]["profile"]["pressure"][::-1].copy()
    assert fastsetup.setup(c4.inputs, opa, wno) is None        # not increasing: the mirror's level loops
    c5 = _case(rng, 31, 1.0, ["H2", "He", "H2O"])
    c5.inputs["atmosphere"]["exclude_mol"] = {"H2O": 0.0}
    assert fastsetup.setup(c5.inputs, opa, wno) is None
    monkeypatch.setattr(opa, "query_method", "nearest")
    assert fastsetup.setup(case.inputs, opa, wno) is None


def test_setup_struct_layout():
    import ctypes
    lib = _lib.load()
    lib.picaso_host_setup_abi.restype = ctypes.c_size_t
    assert lib.picaso_host_setup_abi() == ctypes.sizeof(fastsetup.SetupArgs)


@pytest.mark.parametrize("planet", [False, True])
@pytest.mark.parametrize("seed", range(8))
def test_fast_setup_facets_bits_of_the_mirror(opa, monkeypatch, seed, planet):
    """The facet form of the 3-D path: (nlevel, nfacets) temperature columns, mixing ratios shared or per facet -- the
    facet-form ATMSETUP and the tall plan / coefficients of optics.gas_stage_facets, array by array; with constant gravity
    and (round 5) with a planet radius and mass, gravity G M / z^2 level by level in every facet."""
    import types
    rng = np.random.default_rng(300 + seed)
    wno = opa._wno_test
    nlevel, nfac = int(rng.choice([3, 10, 31])), int(rng.choice([1, 4, 9]))
    cols = ["H2", "He", "H2O", "CH4", "Na"][:int(rng.integers(3, 6))]
    case = _case(rng, nlevel, float(rng.choice([1.0, 1e-9, 1e5, 0.03])), cols, planet=planet)
    inp = case.inputs
    base = inp["atmosphere"]["profile"]
    prof_f = {"pressure": np.asarray(base["pressure"]).reshape(nlevel, 1),
              "temperature": np.ascontiguousarray(np.asarray(base["temperature"])[:, None] * (1.0 + 0.1 * rng.random((1, nfac))))}
    for k in cols:
        v = np.asarray(base[k]).reshape(nlevel, 1)
        prof_f[k] = v * (1.0 + 0.2 * rng.random((1, nfac))) if (seed % 2 and k != "H2") else v
    fast = jdi._setup_atmosphere(inp, opa, wno, prof_f, None)
    assert getattr(fast, "_fast_tall", None) is not None
    monkeypatch.setenv("PICASO_AMD_PY_SETUP", "1")
    ref = jdi._setup_atmosphere(inp, opa, wno, prof_f, None)
    assert getattr(ref, "_fast_tall", None) is None
    nl = nlevel - 1

    def flat(a):
        return np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=float), (nl, nfac)).T).ravel()
    tall = types.SimpleNamespace(c=types.SimpleNamespace(nlayer=nfac * nl, pconv=ref.c.pconv),
                                 layer={"temperature": flat(ref.layer["temperature"]), "pressure": flat(ref.layer["pressure"])},
                                 molecules=ref.molecules, continuum_molecules=ref.continuum_molecules)
    opa.get_opacities(tall, exclude_mol=1)
    pr, fr = opa._plan, px._layer_factors(ref, opa)
    monkeypatch.delenv("PICASO_AMD_PY_SETUP")
    pf, ff = fast._fast_tall[0], fast._fast_tall[1]
    for d in ("level", "layer"):
        fd, rd = getattr(fast, d), getattr(ref, d)
        assert set(fd) == set(rd), (d, set(fd) ^ set(rd))
        for k in rd:
            if k == "mixingratios":
                for m in rd[k]:
                    _same(fd[k][m], rd[k][m], (d, k, m))
            elif k != "cloud":
                _same(fd[k], rd[k], (d, k))
    assert list(fast.molecules) == list(ref.molecules) and fast.continuum_molecules == ref.continuum_molecules
    assert fast.rayleigh_molecules == ref.rayleigh_molecules and fast.warnings == ref.warnings
    assert pf["molecules"] == pr["molecules"] and pf["cia_pairs"] == pr["cia_pairs"] and pf["nlayer"] == pr["nlayer"]
    for k in ("rows", "wts", "fac", "cia_rows"):
        _same(pf[k], pr[k], ("plan", k))
    assert ff[2] == fr[2]
    for i in (0, 1, 3):
        _same(ff[i], fr[i], ("factors", i))


@pytest.fixture()
def opk(monkeypatch):
    """A premixed correlated-k opacity object (RetrieveCKs) on a ragged (P, T) grid, device tables as stand-ins."""
    monkeypatch.setattr(px, "DeviceArray", _Dev)
    nb, nk = 7, 4
    wck = np.linspace(40.0, 28000.0, nb)
    tk = np.array([100.0, 300.0, 700.0, 1500.0, 3000.0])
    pk = np.array([1e-6, 1e-4, 1e-2, 1e-1, 1.0, 10.0, 100.0, 500.0])
    nc_p = np.array([8, 8, 7, 6, 8])                                   # ragged: fewer pressures at some temperatures
    lnk = np.zeros((pk.size, tk.size, nb, nk))
    cia_t = [75.0, 200.0, 500.0, 1000.0, 2000.0, 4000.0]
    cont = {pr: {t: np.ones(nb) for t in cia_t} for pr in ("H2H2", "H2He", "H2CH4")}
    o = px.RetrieveCKs(wck, np.full(nk, 1.0 / nk), np.tile(pk, tk.size), np.repeat(tk, pk.size), nc_p, lnk, continuum=cont,
                       cia_temps=cia_t, rayleigh_opa={m: np.ones(nb) for m in ("H2", "He")}, gauss_pts=np.linspace(0.1, 0.9, nk),
                       ctx=object())
    o._wno_test = wck
    return o


def _compare(f, r, pf, pr, ff, fr, facets=False):
    for d in ("level", "layer"):
        fd, rd = getattr(f, d), getattr(r, d)
        assert set(fd) == set(rd), (d, set(fd) ^ set(rd))
        for k in rd:
            if k == "mixingratios":
                for m in rd[k]:
                    _same(fd[k][m], rd[k][m], (d, k, m))
            elif k != "cloud":
                _same(fd[k], rd[k], (d, k))
    assert list(f.molecules) == list(r.molecules)
    assert f.continuum_molecules == r.continuum_molecules and f.rayleigh_molecules == r.rayleigh_molecules
    assert f.warnings == r.warnings
    assert pf["premixed"] and pr["premixed"] and pf["molecules"] == pr["molecules"] and pf["cia_pairs"] == pr["cia_pairs"]
    assert pf["nlayer"] == pr["nlayer"]
    for k in ("rows", "wts", "fac", "cia_rows", "cia_wts"):
        _same(pf[k], pr[k], ("plan", k))
    assert ff[2] == fr[2]
    for i in (0, 1, 3):
        _same(ff[i], fr[i], ("factors", i))


@pytest.mark.parametrize("seed", range(12))
def test_fast_setup_premixed_k_tables_bits_of_the_mirror(opk, monkeypatch, seed):
    """Round 5: premixed correlated-k tables through the C set-up -- the ragged-grid search of get_mixing_indices with the
    k-table's row numbering, the bracketing continuum temperatures and their 1/T weights (RetrieveCKs._plan_continuum),
    mol_fac = colden / mmw -- against the numpy mirror, array by array."""
    rng = np.random.default_rng(700 + seed)
    cols = ["H2", "He", "H2O", "CH4", "Na"][:int(rng.integers(3, 6))]
    nlevel = int(rng.choice([2, 3, 10, 61, 91]))
    case = _case(rng, nlevel, float(rng.choice([1.0, 1e-9, 1e5, 0.03])), cols, planet=seed % 2 == 1)
    wno = opk._wno_test
    fast = jdi._setup_atmosphere(case.inputs, opk, wno)
    assert getattr(fast, "_fast", None) is not None, "the C set-up declined a k-table profile inside its scope"
    opk.get_opacities(fast)
    pf, ff = opk._plan, px._layer_factors(fast, opk)
    assert pf is fast._fast[0]
    monkeypatch.setenv("PICASO_AMD_PY_SETUP", "1")
    ref = jdi._setup_atmosphere(case.inputs, opk, wno)
    assert getattr(ref, "_fast", None) is None
    opk.get_opacities(ref)
    pr, fr = opk._plan, px._layer_factors(ref, opk)
    monkeypatch.delenv("PICASO_AMD_PY_SETUP")
    _compare(fast, ref, pf, pr, ff, fr)


@pytest.mark.parametrize("seed", range(6))
def test_fast_setup_premixed_k_tables_facet_form(opk, monkeypatch, seed):
    import types
    rng = np.random.default_rng(900 + seed)
    wno = opk._wno_test
    nlevel, nfac = int(rng.choice([3, 10, 31])), int(rng.choice([1, 4, 9]))
    cols = ["H2", "He", "H2O", "CH4"][:int(rng.integers(3, 5))]
    case = _case(rng, nlevel, float(rng.choice([1.0, 1e-9, 0.03])), cols, planet=seed % 2 == 1)
    inp = case.inputs
    base = inp["atmosphere"]["profile"]
    prof_f = {"pressure": np.asarray(base["pressure"]).reshape(nlevel, 1),
              "temperature": np.ascontiguousarray(np.asarray(base["temperature"])[:, None] * (1.0 + 0.1 * rng.random((1, nfac))))}
    for k in cols:
        v = np.asarray(base[k]).reshape(nlevel, 1)
        prof_f[k] = v * (1.0 + 0.2 * rng.random((1, nfac))) if (seed % 2 and k != "H2") else v
    fast = jdi._setup_atmosphere(inp, opk, wno, prof_f, None)
    assert getattr(fast, "_fast_tall", None) is not None
    monkeypatch.setenv("PICASO_AMD_PY_SETUP", "1")
    ref = jdi._setup_atmosphere(inp, opk, wno, prof_f, None)
    assert getattr(ref, "_fast_tall", None) is None
    nl = nlevel - 1

    def flat(a):
        return np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=float), (nl, nfac)).T).ravel()
    tall = types.SimpleNamespace(c=types.SimpleNamespace(nlayer=nfac * nl, pconv=ref.c.pconv),
                                 layer={"temperature": flat(ref.layer["temperature"]), "pressure": flat(ref.layer["pressure"])},
                                 molecules=ref.molecules, continuum_molecules=ref.continuum_molecules)
    opk.get_opacities(tall)
    pr, fr = opk._plan, px._layer_factors(ref, opk)
    monkeypatch.delenv("PICASO_AMD_PY_SETUP")
    _compare(fast, ref, fast._fast_tall[0], pr, fast._fast_tall[1], fr, facets=True)
