"""Cloud tables on their own wavenumber grid (virga's and the box clouds' 196 points): the reference regrids them
row by row with numpy.interp on the host (wavelength.py:46-70 from atmsetup.py:609-622); here ``layer['cloud']``
keeps the compact tables (``atmsetup.CloudTables``) and ``compute_opacity`` regrids on the device
(``picaso_regrid_rows_dev``) -- bit for bit the same numbers."""
import os

import numpy as np
import pytest

from helpers import GOLDEN
from picaso_amd import justdoit as jdi
from picaso_amd.atmsetup import CloudTables

DB = os.path.join(GOLDEN, "synthetic_opacities.db")


def _rows_interp(x, xp, fp):
    return np.stack([np.interp(x, xp, row) for row in fp])


def test_cloud_tables_form_the_reference_arrays_when_read():
    rng = np.random.default_rng(5)
    xp = np.sort(rng.uniform(40.0, 33000.0, 196))
    wno = np.linspace(30.0, 34000.0, 1500)                      # reaches past both ends of the table's grid
    compact = {k: rng.random((12, 196)) for k in ("opd", "w0", "g0")}
    t = CloudTables(compact, xp, wno)
    assert len(t) == 3 and "w0" in t and "tau" not in t and sorted(t.keys()) == ["g0", "opd", "w0"]
    assert not dict.__len__(t)                                   # nothing formed yet
    assert np.array_equal(t["opd"], _rows_interp(wno, xp, compact["opd"])) and t["opd"].flags.c_contiguous
    assert dict.__len__(t) == 1
    blk = t.columns(100, 900)
    for k, v in blk.items():
        assert np.array_equal(v, _rows_interp(wno, xp, compact[k])[:, 100:900])
    with pytest.raises(KeyError):
        t["tau"]
    assert {k: v.shape for k, v in t.items()} == {k: (12, 1500) for k in ("opd", "w0", "g0")}


def _case(og, wgrid=None, holes=False, nwno=None):
    case = jdi.inputs()
    case.phase_angle(0)
    case.gravity(radius=7.1e9, mass=1.9e30)
    prof = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"]}
    for k in ("H2", "He", "H2O", "CH4"):
        prof[k] = og["in/mix/" + k]
    case.atmosphere(df=prof)
    nw = len(og["in/wno"])
    case.star(relative_flux=1.0 + 0.2 * np.cos(np.arange(nw) / 5.0), radius=6.9e10, semi_major=7.5e12)
    case.approx(raman="none")
    return case


def _grid(tmp_path, monkeypatch, lo, hi):
    d = tmp_path / "opacities"
    d.mkdir()
    wn = np.round(np.linspace(lo, hi, 196)[::-1], 2)
    with open(d / "wave_EGP.dat", "w") as fh:
        fh.write("   i   micron.    wavenumber idum     idum1    idum2     idum3\n")
        for i, w in enumerate(wn):
            fh.write("%4d %9.3f %9.2f %8.2f- %7.2f %9.3f %9.3f\n" % (i + 1, 1e4 / w, w, w - 1, w + 1, 2.0, w))
    monkeypatch.setenv("picaso_refdata", str(tmp_path))


def test_atmosphere_keeps_tables_on_their_own_grid_compact(tmp_path, monkeypatch):
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    wno = og["in/wno"]
    _grid(tmp_path, monkeypatch, wno.min() * 1.02, wno.max() * 0.97)         # the opacity grid reaches past it
    case = _case(og)
    case.clouds(g0=[0.8], w0=[0.95], opd=[1.5], p=[0.0], dp=[1.5])
    from picaso_amd.atmsetup import ATMSETUP
    atm = ATMSETUP(case.inputs)
    atm.get_profile()
    atm.get_clouds(wno)
    cld = atm.layer["cloud"]
    assert isinstance(cld, CloudTables) and cld.compact["opd"].shape == (atm.c.nlayer, 196) and not atm.cloud_free
    prof = case.inputs["clouds"]["profile"]
    assert np.array_equal(cld["opd"], _rows_interp(wno, case.inputs["clouds"]["wavenumber"], prof["opd"]))
    assert cld["opd"].max() == 1.5 and cld["opd"].min() == 0.0
    # a table already on the opacity grid stays a plain dict of (nlayer, nwno) arrays
    full = {k: np.array(cld[k]) for k in ("opd", "w0", "g0")}
    case.clouds(df=full)
    atm.input = case.inputs
    atm.get_clouds(wno)
    assert type(atm.layer["cloud"]) is dict and np.array_equal(atm.layer["cloud"]["w0"], full["w0"])


@pytest.mark.gpu
@pytest.mark.parametrize("nin,nwno,nrows", [(196, 5000, 9), (2, 777, 3), (5000, 3001, 2), (196, 100000, 4)])
def test_gpu_regrid_rows_is_numpy_interp(nin, nwno, nrows):
    from picaso_amd import _lib, device
    rng = np.random.default_rng(nin + nwno)
    xp = np.sort(rng.uniform(50.0, 30000.0, nin))
    x = np.sort(rng.uniform(10.0, 31000.0, nwno))                # some columns left and right of the table's grid
    x[rng.integers(0, nwno, 40)] = xp[rng.integers(0, nin, 40)]  # exact knots, first and last among them
    x[:2] = xp[0], xp[-1]
    fp = rng.random((nrows, nin)) * 10.0 ** rng.integers(-8, 3, (nrows, nin))
    fp[0] = 0.3                                                  # a box cloud: constant along the row
    fp[-1, nin // 2:] = 0.0
    ctx = _lib.context()
    d_x = device.DeviceArray.from_host(x, ctx)
    got = device.regrid_rows(xp, fp, d_x, ctx).to_host()
    assert np.array_equal(got, _rows_interp(x, xp, fp))
    got = device.regrid_rows(xp, fp, d_x, ctx, scale=0.37).to_host()
    assert np.array_equal(got, 0.37 * _rows_interp(x, xp, fp))


@pytest.mark.gpu
def test_gpu_regrid_rows_nan_and_inf_follow_numpy():
    from picaso_amd import _lib, device
    xp = np.array([1.0, 2.0, 3.0, 4.0, 5.0])
    fp = np.array([[1.0, np.inf, 3.0, np.nan, np.nan], [2.0, 2.0, np.inf, np.inf, 1.0]])
    x = np.array([0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0, 4.5, 5.0, 6.0, np.nan])
    ctx = _lib.context()
    with np.errstate(invalid="ignore"):
        want = _rows_interp(x, xp, fp)
    got = device.regrid_rows(xp, fp, device.DeviceArray.from_host(x, ctx), ctx).to_host()
    assert np.array_equal(got, want, equal_nan=True)


def _same(a, b, path=""):
    if isinstance(a, dict):
        assert sorted(a.keys()) == sorted(b.keys()), path
        for k in a:
            _same(a[k], b[k], path + "/" + str(k))
    elif isinstance(a, np.ndarray):
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), path
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for x, y in zip(a, b):
            _same(x, y, path)
    else:
        assert a == b or (a != a and b != b), (path, a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("holes,devices", [(False, None), (True, None), (False, [0, 0, 0])])
def test_gpu_box_cloud_spectrum_regridded_on_the_device(tmp_path, monkeypatch, holes, devices):
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    wno = og["in/wno"]
    _grid(tmp_path, monkeypatch, wno.min() * 1.02, wno.max() * 0.97)
    opa = jdi.opannection(filename_db=DB)

    def run():
        case = _case(og)
        case.clouds(g0=[0.8, 0.3], w0=[0.95, 0.7], opd=[1.5, 0.2], p=[0.0, -2.0], dp=[1.5, 1.0],
                    do_holes=holes, fhole=0.3 if holes else None, fthin_cld=0.25 if holes else None)
        return case.spectrum(opa, calculation="reflected+thermal+transmission", full_output=True, devices=devices)
    monkeypatch.delenv("PICASO_AMD_HOST_REGRID", raising=False)
    dev = run()
    monkeypatch.setenv("PICASO_AMD_HOST_REGRID", "1")
    host = run()
    _same(host, dev)
    assert np.isfinite(dev["albedo"]).all() and dev["full_output"]["taucld"].max() > 0
    # and against the table expanded by hand on the host, passed in on the opacity grid
    case = _case(og)
    case.clouds(g0=[0.8, 0.3], w0=[0.95, 0.7], opd=[1.5, 0.2], p=[0.0, -2.0], dp=[1.5, 1.0])
    prof, wg = case.inputs["clouds"]["profile"], case.inputs["clouds"]["wavenumber"]
    case.clouds(df={k: _rows_interp(wno, wg, prof[k]) for k in ("opd", "w0", "g0")}, do_holes=holes,
                fhole=0.3 if holes else None, fthin_cld=0.25 if holes else None)
    monkeypatch.delenv("PICASO_AMD_HOST_REGRID", raising=False)
    byhand = case.spectrum(opa, calculation="reflected+thermal+transmission", devices=devices)
    for k in ("albedo", "thermal", "transit_depth"):
        assert np.array_equal(byhand[k], dev[k]), k


@pytest.mark.gpu
def test_gpu_new_entry_points_refuse_bad_arguments():
    """picaso_regrid_rows_dev / picaso_raman_oklopcic_dev / raman_rows of picaso_compute_opacity_ck_dev: clean error
    codes with a message, nothing launched."""
    import ctypes
    from picaso_amd import _lib, device
    from picaso_amd._lib import PicasoHipError, check, load
    ctx = _lib.context()
    d = device.DeviceArray.from_host(np.arange(8.0), ctx)
    out = device.DeviceArray((2, 8), ctx)
    ci, cl, vp = ctypes.c_int, ctypes.c_long, ctypes.c_void_p
    with pytest.raises(PicasoHipError, match="regrid_rows"):          # a one-point table has no bracket
        check(load().picaso_regrid_rows_dev(ctx, ci(2), ci(1), cl(8), vp(d.addr), vp(d.addr), vp(d.addr), None, vp(out.addr)), ctx)
    with pytest.raises(PicasoHipError, match="regrid_rows"):
        check(load().picaso_regrid_rows_dev(ctx, ci(2), ci(4), cl(8), None, vp(d.addr), vp(d.addr), None, vp(out.addr)), ctx)
    ji = np.array([0, 12], dtype=np.int32)
    isr = np.array([1, 0], dtype=np.int32)
    jat = np.zeros((10, 2))
    with pytest.raises(PicasoHipError, match="j_initial"):
        check(load().picaso_raman_oklopcic_dev(ctx, ci(2), cl(8), ci(2), vp(out.addr), vp(out.addr),
                                               ji.ctypes.data_as(vp), isr.ctypes.data_as(vp), jat.ctypes.data_as(vp),
                                               ctypes.c_double(0.99999), vp(out.addr)), ctx)
    with pytest.raises(PicasoHipError, match="at most"):
        check(load().picaso_raman_oklopcic_dev(ctx, ci(2), cl(8), ci(5000), vp(out.addr), vp(out.addr),
                                               ji.ctypes.data_as(vp), isr.ctypes.data_as(vp), jat.ctypes.data_as(vp),
                                               ctypes.c_double(0.99999), vp(out.addr)), ctx)
    planes = [device.DeviceArray((2, 8), ctx) for _ in range(2)]
    outs = [vp(device.DeviceArray((3 if k in (1, 8) else 2, 8), ctx).addr) for k in range(13)]
    with pytest.raises(PicasoHipError, match="raman_rows"):
        check(load().picaso_compute_opacity_ck_dev(ctx, ci(2), ci(8), ci(1), vp(planes[0].addr), vp(planes[1].addr), None,
                                                   None, None, vp(d.addr), ci(7), ctypes.c_double(0.99999), ci(0), ci(1),
                                                   ci(2), *outs), ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("per_facet", [False, True])
def test_gpu_3d_cloud_tables_on_their_own_grid_regrid_on_the_device(monkeypatch, per_facet):
    """spectrum(dimension='3d') with clouds_3d tables on a wavenumber grid of their own (decreasing, reaching past the
    opacity grid on one side): regridded on the device (picaso_regrid_rows_dev / picaso_regrid_facets_dev) -- the same
    bits as handing in the numpy.interp rows the reference forms facet by facet (atmsetup.py:609-622), and the host
    form of the interpolation (PICASO_AMD_HOST_REGRID=1) to rounding."""
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    ng, nt = 3, 2
    nlayer, nin = len(og["in/tlevel"]) - 1, 9
    rng = np.random.default_rng(3 + per_facet)
    wn = np.linspace(opa.wno[-1] * 0.9, opa.wno[0] * 1.05, nin)[::-1].copy()          # decreasing
    shape = (nlayer, nin, ng, nt) if per_facet else (nlayer, nin)
    cld = {"opd": 0.3 * rng.random(shape), "w0": 0.5 + 0.49 * rng.random(shape), "g0": 0.8 * rng.random(shape)}
    for k in ("opd", "g0"):
        cld[k][:5] = 0.0

    def case(tables):
        c = jdi.inputs()
        c.phase_angle(np.pi / 4, num_gangle=ng, num_tangle=nt)
        c.gravity(gravity=float(og["in/gravity"]))
        prof = {"pressure": og["in/plevel_bar"],
                "temperature": og["in/tlevel"][:, None, None] * (1.0 + 0.02 * np.arange(ng * nt).reshape(1, ng, nt))}
        for k in ("H2", "He", "H2O", "CH4"):
            prof[k] = og["in/mix/" + k]
        c.atmosphere_3d(prof)
        c.clouds_3d(tables)
        c.approx(raman="none")
        return c
    calc = "reflected+thermal"
    monkeypatch.delenv("PICASO_AMD_HOST_REGRID", raising=False)
    dev = case(dict(cld, wavenumber=wn)).spectrum(opa, calculation=calc, dimension="3d")
    # numpy.interp rows on the opacity grid, handed in as arrays already on it
    o = np.argsort(wn)
    on_grid = {}
    for k, v in cld.items():
        rows = np.moveaxis(v.reshape(nlayer, nin, -1), 1, 2)[..., o]                 # (nlayer, nfac|1, nin)
        r = np.stack([[np.interp(opa.wno, wn[o], row) for row in lay] for lay in rows])   # (nlayer, nfac|1, nwno)
        r = np.moveaxis(r, 1, 2)
        on_grid[k] = np.ascontiguousarray(r.reshape(nlayer, opa.nwno, ng, nt) if per_facet else r[:, :, 0])
    ref = case(on_grid).spectrum(opa, calculation=calc, dimension="3d")
    for key in ("albedo", "thermal"):
        assert np.isfinite(dev[key]).all() and np.array_equal(dev[key], ref[key]), key
    # ... which took the facet-major planes of the fused gas + mixing launch (tables interpolated inside it); the
    # facet-fastest planes of round 3's mixing launch (tables regridded and tiled first): the same bits
    from picaso_amd import optics as px
    calls = []
    real = px.compute_opacity_facet_major
    monkeypatch.setattr(px, "compute_opacity_facet_major", lambda *a, **k: (calls.append(k.get("cloud_tables") is not None),
                                                                            real(*a, **k))[1])
    monkeypatch.setenv("PICASO_AMD_NO_DRIVER", "1")       # the call-by-call path (the C driver makes the same launches itself)
    again = case(dict(cld, wavenumber=wn)).spectrum(opa, calculation=calc, dimension="3d")
    assert calls == [True]
    monkeypatch.setenv("PICASO_AMD_FACET_FASTEST", "1")
    ff = case(dict(cld, wavenumber=wn)).spectrum(opa, calculation=calc, dimension="3d")
    assert calls == [True]
    monkeypatch.delenv("PICASO_AMD_FACET_FASTEST")
    monkeypatch.delenv("PICASO_AMD_NO_DRIVER")
    for key in ("albedo", "thermal"):
        assert np.array_equal(again[key], dev[key]) and np.array_equal(ff[key], dev[key]), key
    for one in ("reflected", "thermal"):
        a = case(dict(cld, wavenumber=wn)).spectrum(opa, calculation=one, dimension="3d")
        assert np.array_equal(a["albedo" if one == "reflected" else "thermal"], dev["albedo" if one == "reflected" else "thermal"])
    monkeypatch.setenv("PICASO_AMD_HOST_REGRID", "1")
    host = case(dict(cld, wavenumber=wn)).spectrum(opa, calculation=calc, dimension="3d")
    for key in ("albedo", "thermal"):
        assert np.max(np.abs(host[key] - dev[key]) / np.abs(dev[key])) < 1e-11, key


@pytest.mark.gpu
def test_gpu_3d_cloud_tables_edited_in_place_are_seen():
    """The compact tables stay resident with the cloud dictionary between calls; scaling an array in place (same
    object) must drop them: the second spectrum equals one from a fresh dictionary with the scaled values."""
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    ng, nt = 2, 2
    nlayer, nin = len(og["in/tlevel"]) - 1, 6
    rng = np.random.default_rng(12)
    wn = np.linspace(opa.wno[0], opa.wno[-1], nin)
    cld = {"opd": 0.2 * rng.random((nlayer, nin, ng, nt)), "w0": 0.5 + 0.4 * rng.random((nlayer, nin, ng, nt)),
           "g0": 0.6 * rng.random((nlayer, nin, ng, nt)), "wavenumber": wn}

    def run(tables):
        c = jdi.inputs()
        c.phase_angle(0.5, num_gangle=ng, num_tangle=nt)
        c.gravity(gravity=float(og["in/gravity"]))
        prof = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"][:, None, None] * np.ones((1, ng, nt))}
        for k in ("H2", "He", "H2O", "CH4"):
            prof[k] = og["in/mix/" + k]
        c.atmosphere_3d(prof)
        c.approx(raman="none")
        c.inputs["clouds"]["profile_3d"] = tables            # the dictionary itself, as phase_curve(clouds_by_phase=) hands it on
        c.inputs["clouds"]["dims"] = "3d"
        return c.spectrum(opa, calculation="reflected", dimension="3d")["albedo"]
    first = run(cld)
    assert set(cld) == {"opd", "w0", "g0", "wavenumber"}          # the caller's dictionary is not written to
    import copy
    assert np.array_equal(run(copy.deepcopy(cld)), first)
    cld["opd"] *= 3.0
    second = run(cld)
    fresh = run({k: (v.copy() if k != "wavenumber" else v) for k, v in cld.items() if not k.startswith("_")})
    assert np.array_equal(second, fresh) and not np.array_equal(second, first)
    # one (layer, facet) row rewritten, then a single element changed: every edit reaches the device (the tables are keyed
    # by a digest of all their bytes; round 4's strided sample missed 29 % of the row edits and every single-element one)
    last = second
    for trial in range(12):
        lay, g, t = int(rng.integers(4)), int(rng.integers(ng)), int(rng.integers(nt))     # near the top: always visible
        if trial % 2 == 0:
            cld["opd"][lay, :, g, t] = 0.2 + rng.random(nin)
        else:
            cld["opd"][lay, int(rng.integers(nin)), g, t] += 0.3
        got = run(cld)
        fresh = run({k: v.copy() for k, v in cld.items()})
        assert np.array_equal(got, fresh) and not np.array_equal(got, last), trial
        last = got
