"""Correlated-k table FILES -> arrays -> HBM-resident tables (reference
``get_ck_tables`` opacity_factory.py:2221-2327, ``RetrieveCKs.__init__`` / ``get_h5_data`` optics.py:676-770,
``opannection`` justdoit.py:1296-1419).  The files are written here in the reference's layouts: per-gas
``<gas>_1460.npy`` directories (axes from ``$picaso_refdata``), per-gas and premixed HDF5 (through h5py where it is
installed, else through the stand-in of tests/helpers.py registered under that name: the reader branch always runs),
the sqlite continuum database."""
import os
import sqlite3
import io

import numpy as np
import pytest

from helpers import GOLDEN

DB = os.path.join(GOLDEN, "synthetic_opacities.db")
NC_P = np.array([6, 6, 5, 6, 4])           # ragged: table points per temperature
TEMPS = np.array([200.0, 400.0, 700.0, 1100.0, 1600.0])
PRESS = np.array([1e-6, 1e-4, 1e-2, 1.0, 30.0, 100.0])


def _grid_rows():
    rows = []
    for t, n in zip(TEMPS, NC_P):
        for p in PRESS[:n]:
            rows.append((t, p))
    return rows


def _refdata(tmp_path, wno):
    (tmp_path / "climate_INPUTS").mkdir()
    (tmp_path / "opacities").mkdir()
    dw = np.gradient(wno)
    np.savetxt(tmp_path / "climate_INPUTS" / "wvno_661", np.column_stack([wno, dw]))
    with open(tmp_path / "opacities" / "grid1460.csv", "w") as fh:
        fh.write("file_number,temperature_K,pressure_bar,number_wave_pts,delta_wavenumber,start_wavenumber\n")
        for i, (t, p) in enumerate(_grid_rows()):
            fh.write("%d,%r,%r,10921760,0.0035,30.0\n" % (i + 1, float(t), float(p)))
    return str(tmp_path), dw


def _tables(nwno, ngauss, seed=0):
    rng = np.random.default_rng(seed)
    out = {}
    for m in ("H2O", "CH4", "H2"):
        base = rng.uniform(-70.0, -45.0, (PRESS.size, TEMPS.size, nwno, 1))
        out[m] = base + np.sort(rng.uniform(0.0, 6.0, (PRESS.size, TEMPS.size, nwno, ngauss)), axis=-1)   # ln kappa
    return out


def test_g_w_2gauss():
    from picaso_amd import optics as px
    g, w = px.g_w_2gauss(order=4, gfrac=0.95)
    assert g.shape == w.shape == (8,)
    assert np.isclose(w.sum(), 1.0) and np.isclose(w[:4].sum(), 0.95)
    assert np.all(np.diff(g) > 0) and g[3] < 0.95 < g[4]
    x, wx = np.polynomial.legendre.leggauss(4)
    assert np.array_equal(g[:4], 0.95 * 0.5 * (x + 1.0)) and np.array_equal(w[4:], (1.0 - 0.95) * wx * 0.5)


def test_read_npy_directory(tmp_path):
    from picaso_amd import optics as px
    wno = np.linspace(50.0, 30000.0, 23)
    ref, dw = _refdata(tmp_path, wno)
    d = tmp_path / "resortrebin"
    d.mkdir()
    tabs = _tables(wno.size, 8)
    for m, a in tabs.items():
        np.save(d / ("%s_1460.npy" % m), a)
    t = px.read_ck_tables(str(d), preload_gases="all", refdata=ref)
    assert sorted(t["molecules"]) == ["CH4", "H2", "H2O"]
    for m in tabs:
        assert np.array_equal(t["kappas"][m], tabs[m])
    assert np.allclose(t["wno"], wno) and np.allclose(t["delta_wno"], dw)
    assert np.array_equal(t["pressures"], PRESS) and np.array_equal(t["temps"], TEMPS)
    assert np.array_equal(t["nc_p"], NC_P)
    g, w = px.g_w_2gauss()
    assert np.array_equal(t["gauss_pts"], g) and np.array_equal(t["gauss_wts"], w)
    one = px.read_ck_tables(str(d), preload_gases=["CH4"], refdata=ref)
    assert one["molecules"] == ["CH4"]
    with pytest.raises(Exception, match="No .npy or .hdf5"):
        (tmp_path / "empty").mkdir()
        px.read_ck_tables(str(tmp_path / "empty"), preload_gases="all", refdata=ref)
    with pytest.raises(Exception, match="does not exist"):
        px.read_ck_tables(str(tmp_path / "nowhere"))


def test_read_continuum_db():
    from picaso_amd import optics as px
    wno, cont, temps = px.read_continuum_db(DB)
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    assert np.array_equal(wno, og["in/wno"])
    assert {"H2H2", "H2He", "H2CH4"} <= set(cont)
    assert all(v.shape == wno.shape for d in cont.values() for v in d.values())
    assert np.array_equal(temps, np.array(sorted({t for d in cont.values() for t in d})))
    with pytest.raises(Exception, match="does not exist"):
        px.read_continuum_db("/nonexistent/cont.db")


def _write_h5(path, wno, dw, kcoeffs, gauss, premixed):
    import h5py                       # the `h5py` fixture has registered the package or its stand-in
    rows = _grid_rows()
    with h5py.File(path, "w") as f:
        f["wno"], f["delta_wno"] = wno, dw
        f["pressures"] = np.array([p for _, p in rows])
        f["temperatures"] = np.array([t for t, _ in rows])
        f["gauss_pts"], f["gauss_wts"] = gauss
        f["kcoeffs"] = kcoeffs
        if premixed:
            f["ck_molecules"] = np.array([b"H2O", b"CH4"])
            f["abunds"] = np.arange(len(rows) * 2, dtype=float).reshape(len(rows), 2)
            f["abunds_map"] = np.array([b"H2O", b"CH4"])
        else:
            f["nc_p"] = NC_P


def test_read_hdf5_premixed_and_per_gas(tmp_path, h5py):
    from picaso_amd import optics as px
    wno = np.linspace(50.0, 30000.0, 17)
    dw = np.gradient(wno)
    gauss = px.g_w_2gauss()
    tabs = _tables(wno.size, 8, seed=3)
    pm = str(tmp_path / "premixed_m+0.0_co1.0.hdf5")
    _write_h5(pm, wno, dw, tabs["H2O"], gauss, True)
    t = px.read_ck_tables(pm)
    assert np.array_equal(t["kappa"], tabs["H2O"]) and t["molecules"] == ["H2O", "CH4"]
    assert np.array_equal(t["nc_p"], NC_P) and np.array_equal(t["temps"], TEMPS)
    assert np.array_equal(t["pressures"], PRESS) and np.array_equal(t["gauss_wts"], gauss[1])
    assert np.array_equal(t["wno"], wno) and np.array_equal(t["delta_wno"], dw)
    assert t["abunds_map"] == ["H2O", "CH4"] and t["abunds"].shape == (NC_P.sum(), 2)
    d = tmp_path / "by_molecule"
    d.mkdir()
    for m in tabs:
        _write_h5(str(d / ("%s_1460.hdf5" % m)), wno, dw, tabs[m], gauss, False)
    t = px.read_ck_tables(str(d), preload_gases="all")
    assert sorted(t["molecules"]) == ["CH4", "H2", "H2O"] and np.array_equal(t["kappas"]["CH4"], tabs["CH4"])
    assert np.array_equal(t["nc_p"], NC_P)
    # the same tables through the .npy route (axes from $picaso_refdata): every array bit for bit
    ref, _ = _refdata(tmp_path, wno)
    dn = tmp_path / "by_molecule_npy"
    dn.mkdir()
    for m, a in tabs.items():
        np.save(dn / ("%s_1460.npy" % m), a)
    n = px.read_ck_tables(str(dn), preload_gases="all", refdata=ref)
    assert sorted(n["molecules"]) == sorted(t["molecules"])
    for m in tabs:
        assert np.array_equal(n["kappas"][m], t["kappas"][m])
    for k in ("pressures", "temps", "nc_p", "gauss_pts", "gauss_wts"):
        assert np.array_equal(n[k], t[k]), k
    assert np.allclose(n["wno"], wno, rtol=1e-15) and np.allclose(n["delta_wno"], dw, rtol=1e-15)     # through a text file
    # a requested gas without a table: the reference says so and goes on (opacity_factory.py:2286-2290)
    with pytest.warns(UserWarning, match="no k-table for NH3"):
        t2 = px.read_ck_tables(str(d), preload_gases=["H2O", "NH3"])
    assert t2["molecules"] == ["H2O"]
    with pytest.raises(Exception, match="No molecules are left to mix"), pytest.warns(UserWarning):
        px.read_ck_tables(str(d), preload_gases=["NH3"])


def test_hdf5_without_h5py_is_a_clear_error(tmp_path, monkeypatch):
    import builtins
    from picaso_amd import optics as px
    real = builtins.__import__

    def fake(name, *a, **k):
        if name == "h5py":
            raise ImportError("no h5py")
        return real(name, *a, **k)
    monkeypatch.setattr(builtins, "__import__", fake)
    f = tmp_path / "x.hdf5"
    f.write_bytes(b"")
    with pytest.raises(Exception, match="needs the h5py package"):
        px.read_ck_tables(str(f))


def _cont_db_on(path, wno):
    """A continuum database in the reference schema on the k-table grid (pairs of the synthetic DB, resampled)."""
    from picaso_amd import optics as px
    w0, cont, temps = px.read_continuum_db(DB)

    def blob(a):
        out = io.BytesIO()
        np.save(out, a)
        return sqlite3.Binary(out.getvalue())
    conn = sqlite3.connect(path)
    conn.execute("CREATE TABLE header (id INTEGER PRIMARY KEY, pressure_unit VARCHAR, temperature_unit VARCHAR, "
                 "wavenumber_grid array, continuum_unit VARCHAR, molecular_unit VARCHAR)")
    conn.execute("CREATE TABLE continuum (id INTEGER PRIMARY KEY, molecule VARCHAR, temperature FLOAT, opacity array)")
    conn.execute("INSERT INTO header (pressure_unit, temperature_unit, wavenumber_grid, continuum_unit, molecular_unit) "
                 "VALUES (?,?,?,?,?)", ("bar", "kelvin", blob(wno), "cm-1 amagat-2", "cm2/molecule"))
    order = np.argsort(w0)
    for mol, d in cont.items():
        for t, k in d.items():
            conn.execute("INSERT INTO continuum (molecule, temperature, opacity) VALUES (?,?,?)",
                         (mol, t, blob(np.interp(wno, w0[order], k[order]))))
    conn.commit()
    conn.close()


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["resortrebin", "resortrebin_hdf5", "preweighted"])
def test_gpu_opannection_from_ck_files_equals_arrays(tmp_path, method, h5py):
    """opannection(method=..., ck_db=<files>, filename_db=<continuum db>) builds the same resident tables as
    RetrieveCKs(<arrays>): a reflected + thermal spectrum through both is bit-identical."""
    from picaso_amd import justdoit as jdi
    from picaso_amd import optics as px
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    wno = np.sort(og["in/wno"])
    ref, dw = _refdata(tmp_path, wno)
    cdb = str(tmp_path / "cont.db")
    _cont_db_on(cdb, wno)
    tabs = _tables(wno.size, 8, seed=11)
    ray = {m: 1e-27 * (wno / 1e4) ** 4 * s for m, s in (("H2", 1.0), ("He", 0.1), ("CH4", 9.0))}
    gauss = px.g_w_2gauss()
    if method == "resortrebin":
        d = tmp_path / "resortrebin"
        d.mkdir()
        for m, a in tabs.items():
            np.save(d / ("%s_1460.npy" % m), a)
        os.environ["picaso_refdata"] = ref
        try:
            opa = jdi.opannection(method="resortrebin", ck_db=str(d), filename_db=cdb, rayleigh_opa=ray)
        finally:
            del os.environ["picaso_refdata"]
        assert opa.on_fly and sorted(opa.preload_gases) == ["CH4", "H2", "H2O"]
    elif method == "resortrebin_hdf5":            # per-gas HDF5 tables: axes from the files themselves
        d = tmp_path / "by_molecule"
        d.mkdir()
        for m, a in tabs.items():
            _write_h5(str(d / ("%s_1460.hdf5" % m)), wno, dw, a, gauss, False)
        opa = jdi.opannection(method="resortrebin", ck_db=str(d), filename_db=cdb, rayleigh_opa=ray)
        assert opa.on_fly and sorted(opa.preload_gases) == ["CH4", "H2", "H2O"]
    else:
        f = str(tmp_path / "pm.hdf5")
        _write_h5(f, wno, dw, tabs["H2O"], gauss, True)
        opa = jdi.opannection(method="preweighted", ck_db=f, filename_db=cdb, rayleigh_opa=ray)
    assert opa.ngauss == 8 and np.allclose(opa.delta_wno, dw)
    _, cont, ctemps = px.read_continuum_db(cdb)
    pressures = np.concatenate([PRESS[:n] for n in NC_P])
    temps_flat = np.concatenate([[t] * n for t, n in zip(TEMPS, NC_P)])
    kw = dict(kappas=tabs, on_fly=True) if method.startswith("resortrebin") else dict(ln_kappa=tabs["H2O"])
    arr = px.RetrieveCKs(wno, gauss[1], pressures, temps_flat, NC_P, continuum=cont, cia_temps=ctemps,
                         rayleigh_opa=ray, gauss_pts=gauss[0], **kw)

    def run(o):
        case = jdi.inputs()
        case.phase_angle(0)
        case.gravity(gravity=float(og["in/gravity"]))
        prof = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"]}
        for k in ("H2", "He", "H2O", "CH4"):
            prof[k] = og["in/mix/" + k]
        case.atmosphere(df=prof)
        case.approx(raman="none")
        return case.spectrum(o, calculation="reflected+thermal")
    a, b = run(opa), run(arr)
    assert np.isfinite(a["albedo"]).all() and np.isfinite(a["thermal"]).all() and a["albedo"].max() > 0
    assert np.array_equal(a["albedo"], b["albedo"]) and np.array_equal(a["thermal"], b["thermal"])
