"""picaso_toon_spectrum_blocks (csrc/driver.hip): one C call enqueues gas stage -> compute_opacity -> reflected ||
thermal for every wavelength block of a 1-D Toon or SH spectrum (reference sequence justdoit.py:236-385; fan-out
justdoit.py:4774).  It only chains the library's entry points, so every output must equal the call-by-call path of
``justdoit.picaso`` (``PICASO_AMD_NO_DRIVER=1``) bit for bit -- single GPU and wavelength blocks, cloud-free (three planes
+ aliases) and cloudy (host cloud planes cut per block inside the C call), Raman off / Pollack, one or both legs, with
and without a star -- and calls it does not cover must fall through unchanged."""
import os

import numpy as np
import pytest

from helpers import GOLDEN
from test_devices_gpu import _same

pytestmark = pytest.mark.gpu
DB = os.path.join(GOLDEN, "synthetic_opacities.db")


def _case(og, jdi, cloud, star, raman, surf_array, k=0):
    case = jdi.inputs(calculation="planet" if star else "browndwarf")
    case.phase_angle(0)
    case.gravity(gravity=float(og["in/gravity"]), radius=7.1e9, mass=1.9e30)
    prof = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"] * (1.0 + 0.01 * k)}
    for m in ("H2", "He", "H2O", "CH4"):
        prof[m] = og["in/mix/" + m]
    case.atmosphere(df=prof)
    if cloud:
        case.clouds(df={"opd": og["in/cld_opd"], "w0": og["in/cld_w0"], "g0": og["in/cld_g0"]})
    nwno = len(og["in/wno"])
    if star:
        case.star(relative_flux=1.0 + 0.3 * np.sin(np.arange(nwno) / 7.0), radius=6.9e10, semi_major=7.5e12)
    case.surface_reflect(0.1 + 0.2 * np.cos(np.arange(nwno) / 11.0) ** 2 if surf_array else 0.15)
    case.approx(raman=raman, delta_eddington=True)
    return case


@pytest.fixture
def pollack_table(tmp_path, monkeypatch):
    d = tmp_path / "opacities"
    d.mkdir()
    wl = np.linspace(0.2, 6.0, 300)
    np.savetxt(d / "raman_fortran.txt", np.column_stack([wl, 0.9 + 0.05 * np.cos(3 * wl)]))
    monkeypatch.setenv("picaso_refdata", str(tmp_path))


@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
@pytest.mark.parametrize("calc", ["reflected", "thermal", "reflected+thermal"])
@pytest.mark.parametrize("cloud,star,raman,surf_array", [(False, True, "none", False), (True, True, "none", True),
                                                         (False, False, "none", True), (True, True, "pollack", False),
                                                         (False, True, "pollack", True)])
def test_driver_equals_call_by_call_path(monkeypatch, pollack_table, devices, calc, cloud, star, raman, surf_array):
    from picaso_amd import justdoit as jdi
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    monkeypatch.setenv("PICASO_AMD_NO_DRIVER", "1")
    want = [_case(og, jdi, cloud, star, raman, surf_array, k).spectrum(opa, calculation=calc, devices=devices) for k in range(2)]
    assert "_driver_tables" not in opa.__dict__
    monkeypatch.delenv("PICASO_AMD_NO_DRIVER")
    got = [_case(og, jdi, cloud, star, raman, surf_array, k).spectrum(opa, calculation=calc, devices=devices) for k in range(2)]
    assert len(opa.__dict__["_driver_tables"]) == 1          # the second spectrum reused the first one's block table
    for w, g in zip(want, got):
        _same(w, g)


@pytest.mark.parametrize("cloud", [False, True])
def test_driver_blocks_enqueued_from_threads(monkeypatch, cloud):
    """Wavelength blocks on different devices are enqueued from a thread each (csrc/driver.hip); on this pool's single
    GPU that form is forced here (PICASO_AMD_PARALLEL_BLOCKS=1) over five blocks, twenty times: the serial loop's bits."""
    from picaso_amd import justdoit as jdi
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    devs = [0, 0, 0, 0, 0]
    monkeypatch.setenv("PICASO_AMD_PARALLEL_BLOCKS", "0")
    want = _case(og, jdi, cloud, True, "none", True).spectrum(opa, calculation="reflected+thermal", devices=devs)
    monkeypatch.setenv("PICASO_AMD_PARALLEL_BLOCKS", "1")
    for _ in range(20):
        _same(want, _case(og, jdi, cloud, True, "none", True).spectrum(opa, calculation="reflected+thermal", devices=devs))
    _same(want, _case(og, jdi, cloud, True, "none", True).spectrum(opa, calculation="reflected+thermal"))


@pytest.mark.parametrize("kind", ["sh", "3d"])
def test_driver_sh_and_3d_blocks_enqueued_from_threads(monkeypatch, kind):
    """The threaded enqueue (blocks on different devices; forced here onto the one GPU) for the round-5 block kinds: SH4
    blocks and 3-D blocks, four blocks, ten times each: the serial loop's bits."""
    from picaso_amd import justdoit as jdi
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    devs = [0, 0, 0, 0]

    def run(devices):
        if kind == "sh":
            c = _case(og, jdi, True, True, "none", True)
            c.approx(raman="none", delta_eddington=True, rt_method="SH", stream=4)
            return c.spectrum(opa, calculation="reflected+thermal", devices=devices)
        c = jdi.inputs()
        c.phase_angle(0.7, num_gangle=3, num_tangle=2)
        c.gravity(gravity=float(og["in/gravity"]))
        prof = {"pressure": og["in/plevel_bar"],
                "temperature": og["in/tlevel"][:, None, None] * (1.0 + 0.02 * np.arange(6).reshape(1, 3, 2))}
        for m in ("H2", "He", "H2O", "CH4"):
            prof[m] = og["in/mix/" + m]
        c.atmosphere_3d(prof)
        c.approx(raman="none")
        return c.spectrum(opa, calculation="reflected+thermal", dimension="3d", devices=devices)
    monkeypatch.setenv("PICASO_AMD_PARALLEL_BLOCKS", "0")
    want = run(devs)
    assert len(opa.__dict__["_driver_tables"]) == 1
    monkeypatch.setenv("PICASO_AMD_PARALLEL_BLOCKS", "1")
    for _ in range(10):
        _same(want, run(devs))
    _same(want, run(None))


def test_driver_nearest_query_and_falls_through_where_it_does_not_apply(monkeypatch):
    from picaso_amd import justdoit as jdi
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    opa = jdi.opannection(filename_db=DB, query_method="nearest")
    got = _case(og, jdi, True, True, "none", True).spectrum(opa, calculation="reflected+thermal")
    assert len(opa.__dict__["_driver_tables"]) == 1
    monkeypatch.setenv("PICASO_AMD_NO_DRIVER", "1")
    _same(_case(og, jdi, True, True, "none", True).spectrum(opa, calculation="reflected+thermal"), got)
    monkeypatch.delenv("PICASO_AMD_NO_DRIVER")
    n = len(opa.__dict__["_driver_tables"])
    # outside the driver: full_output, transmission, SH layer fluxes, level fluxes -- the usual path, no new block table
    _case(og, jdi, True, True, "none", True).spectrum(opa, calculation="reflected", full_output=True)
    _case(og, jdi, False, True, "none", False).spectrum(opa, calculation="reflected+transmission")
    sh = _case(og, jdi, True, True, "none", False)
    sh.approx(raman="none", rt_method="SH", stream=4, calculate_fluxes="on")
    sh.spectrum(opa, calculation="reflected")
    lv = _case(og, jdi, False, True, "none", False)
    lv.approx(raman="none", get_lvl_flux=True)
    lv.spectrum(opa, calculation="reflected+thermal")
    assert len(opa.__dict__["_driver_tables"]) == n


def test_driver_lean_planes(monkeypatch):
    from picaso_amd import resident
    """Through the driver as well a cloud-free atmosphere writes two planes (dtau, w0: the reflected kernel re-derives the
    rest, the thermal one gets aliases and a constant), a cloudy one eight of the eleven (tau, tau_og and gcos2 are
    re-derived); PICASO_AMD_ALL_PLANES=1 writes the full set -- same bits."""
    from picaso_amd import justdoit as jdi
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    lean = _case(og, jdi, False, True, "none", True).spectrum(opa, calculation="reflected+thermal")
    (t_lean,) = opa.__dict__["_driver_tables"].values()
    assert set(t_lean.want) == {"dtau", "w0"}
    monkeypatch.setenv("PICASO_AMD_ALL_PLANES", "1")
    full = _case(og, jdi, False, True, "none", True).spectrum(opa, calculation="reflected+thermal")
    t_full = [t for t in opa.__dict__["_driver_tables"].values() if t is not t_lean][0]
    assert len(t_full.want) == 12                          # the eleven reflected-light planes + w0_no_raman
    _same(full, lean)
    monkeypatch.delenv("PICASO_AMD_ALL_PLANES")
    _case(og, jdi, True, True, "none", True).spectrum(opa, calculation="reflected")
    t_cld = [t for t in opa.__dict__["_driver_tables"].values() if t is not t_lean and t is not t_full][0]
    assert set(t_cld.want) == set(resident.REFLECTED_PLANES) - {"tau", "tau_og", "gcos2"}
    cloudy = _case(og, jdi, True, True, "none", True).spectrum(opa, calculation="reflected+thermal")
    monkeypatch.setenv("PICASO_AMD_ALL_PLANES", "1")
    _same(_case(og, jdi, True, True, "none", True).spectrum(opa, calculation="reflected+thermal"), cloudy)


def test_driver_cloud_tables_on_their_own_grid(monkeypatch):
    """Cloud tables on a wavenumber grid of their own (virga; the box-cloud form of clouds()): regridded on the device and
    handed to the driver as device planes -- same bits as the call-by-call path and as the host regrid."""
    from picaso_amd import justdoit as jdi
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    nlayer = len(og["in/tlevel"]) - 1
    rng = np.random.default_rng(3)
    cw = np.sort(rng.uniform(opa.wno.min() * 0.9, opa.wno.max() * 1.1, 37))

    def make():
        case = _case(og, jdi, False, True, "none", True)
        case.clouds(df={"opd": 0.3 * rng.random((nlayer, 37)), "w0": 0.5 + 0.49 * rng.random((nlayer, 37)),
                        "g0": 0.9 * rng.random((nlayer, 37))}, wavenumber=cw)
        return case
    case = make()
    from picaso_amd.atmsetup import CloudTables
    atm = jdi._setup_atmosphere(case.inputs, opa, opa.wno)
    if not isinstance(atm.layer["cloud"], CloudTables):
        pytest.skip("clouds(df=) with a wavenumber column does not produce tables on their own grid here")
    got = case.spectrum(opa, calculation="reflected+thermal")
    assert len(opa.__dict__.get("_driver_tables", {})) == 1
    monkeypatch.setenv("PICASO_AMD_NO_DRIVER", "1")
    _same(case.spectrum(opa, calculation="reflected+thermal"), got)
    monkeypatch.setenv("PICASO_AMD_HOST_REGRID", "1")
    _same(case.spectrum(opa, calculation="reflected+thermal"), got)


@pytest.mark.parametrize("cloud", [False, True])
@pytest.mark.parametrize("calc", ["reflected", "reflected+thermal"])
def test_driver_oklopcic_raman(monkeypatch, cloud, calc):
    """raman='oklopcic': the factor plane is formed on the device per call (layer temperatures) and handed to the driver's
    opacity launch -- the bits of the call-by-call path."""
    from picaso_amd import justdoit as jdi
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    opa.raman_stellar_shifts = og["in/raman_shifts"]
    opa.raman_db = {"c": og["in/raman_c"], "ji": og["in/raman_ji"], "deltanu": og["in/raman_deltanu"]}
    got = [_case(og, jdi, cloud, True, "oklopcic", True, k).spectrum(opa, calculation=calc) for k in range(2)]
    assert len(opa.__dict__["_driver_tables"]) == 1
    monkeypatch.setenv("PICASO_AMD_NO_DRIVER", "1")
    for k in range(2):
        _same(_case(og, jdi, cloud, True, "oklopcic", True, k).spectrum(opa, calculation=calc), got[k])


@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
@pytest.mark.parametrize("calc", ["reflected", "thermal", "reflected+thermal"])
@pytest.mark.parametrize("cloud,stream,forms", [(False, 4, {}), (True, 4, {}), (True, 2, {}),
                                                (True, 4, dict(w_single_form="OTHG", psingle_form="isotropic",
                                                               single_form="legendre", w_multi_rayleigh="off"))])
def test_driver_runs_the_sh_solvers(monkeypatch, devices, calc, cloud, stream, forms):
    """rt_method='SH' through the C driver (round 5: it used to take the Python per-block loop, 0.2 ms of interpreter time
    per wavelength block): cloud-free (dtau and w0 only), a cloud deck with the default forms (eight planes, level planes
    derived, the layers above the deck through the cloud-free kernel) and other forms (all thirteen planes) -- whole grid
    and wavelength blocks, every output equal to the call-by-call path bit for bit."""
    from picaso_amd import justdoit as jdi
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    opa = jdi.opannection(filename_db=DB, query_method="linear")

    def make(k):
        c = _case(og, jdi, cloud, True, "none", True, k)
        c.approx(raman="none", delta_eddington=True, rt_method="SH", stream=stream, **forms)
        return c
    monkeypatch.setenv("PICASO_AMD_NO_DRIVER", "1")
    want = [make(k).spectrum(opa, calculation=calc, devices=devices) for k in range(2)]
    assert "_driver_tables" not in opa.__dict__
    monkeypatch.delenv("PICASO_AMD_NO_DRIVER")
    got = [make(k).spectrum(opa, calculation=calc, devices=devices) for k in range(2)]
    assert len(opa.__dict__["_driver_tables"]) == 1
    (table,) = opa.__dict__["_driver_tables"].values()
    if not cloud and stream == 4:
        assert set(table.want) == {"dtau", "w0"}
    elif not forms:
        assert set(table.want) == {"dtau", "w0", "cosb_og", "ftau_cld", "ftau_ray", "f_deltaM", "dtau_og", "w0_og"}
    else:
        assert len(table.want) == 13
    for w, g in zip(want, got):
        _same(w, g)
        assert all(np.isfinite(v).all() for v in g.values() if isinstance(v, np.ndarray))


@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
@pytest.mark.parametrize("calc", ["reflected", "thermal", "reflected+thermal"])
@pytest.mark.parametrize("cloud,raman", [(None, "none"), (None, "pollack"), ("shared", "none"), ("per_facet", "none")])
def test_driver_runs_the_3d_blocks(monkeypatch, pollack_table, devices, calc, cloud, raman):
    """dimension='3d' through the C driver (round 5; it took the Python per-block loop, ~0.45 ms of interpreter time per
    wavelength block): ONE fused gas + mixing launch over the tall atmosphere of all facets, the batched 3-D solvers with
    every facet as a spectrum of its own, the disk sums -- cloud-free maps and cloud tables on their own wavenumber grid
    (one table for the disk, one per facet), whole grid and wavelength blocks, every output equal to the call-by-call
    path (spectrum.Spectrum) bit for bit."""
    from picaso_amd import justdoit as jdi
    og = np.load(os.path.join(GOLDEN, "optics.npz"))
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    ng, nt = 3, 2
    nlayer, nin = len(og["in/tlevel"]) - 1, 7
    rng = np.random.default_rng(5)
    wn = np.linspace(opa.wno[-1] * 0.95, opa.wno[0] * 1.02, nin)
    shape = (nlayer, nin, ng, nt) if cloud == "per_facet" else (nlayer, nin)
    cld = {"opd": 0.3 * rng.random(shape), "w0": 0.5 + 0.49 * rng.random(shape), "g0": 0.8 * rng.random(shape), "wavenumber": wn}
    for k in ("opd", "g0"):
        cld[k][:4] = 0.0

    def make(k):
        c = jdi.inputs()
        c.phase_angle(0.6 + 0.2 * k, num_gangle=ng, num_tangle=nt)
        c.gravity(gravity=float(og["in/gravity"]), radius=7.1e9, mass=1.9e30)
        nwno = opa.nwno
        c.star(relative_flux=1.0 + 0.3 * np.sin(np.arange(nwno) / 7.0), radius=6.9e10, semi_major=7.5e12)
        c.surface_reflect(0.1 + 0.2 * np.cos(np.arange(nwno) / 11.0) ** 2)
        prof = {"pressure": og["in/plevel_bar"],
                "temperature": og["in/tlevel"][:, None, None] * (1.0 + 0.015 * (1 + k) * np.arange(ng * nt).reshape(1, ng, nt))}
        for m in ("H2", "He", "H2O", "CH4"):
            prof[m] = og["in/mix/" + m]
        c.atmosphere_3d(prof)
        if cloud:
            c.clouds_3d(dict(cld))
        c.approx(raman=raman)
        return c
    monkeypatch.setenv("PICASO_AMD_NO_DRIVER", "1")
    want = [make(k).spectrum(opa, calculation=calc, dimension="3d", devices=devices) for k in range(2)]
    assert "_driver_tables" not in opa.__dict__
    monkeypatch.delenv("PICASO_AMD_NO_DRIVER")
    got = [make(k).spectrum(opa, calculation=calc, dimension="3d", devices=devices) for k in range(2)]
    assert len(opa.__dict__["_driver_tables"]) == 1          # the second spectrum reused the first one's blocks
    (table,) = opa.__dict__["_driver_tables"].values()
    if cloud is None and "reflected" in calc:
        assert set(table.want) == {"dtau", "w0"} | ({"w0_no_raman"} if ("thermal" in calc and raman != "none") else set())
    assert not ({"tau", "tau_og"} & set(table.want))
    for w, g in zip(want, got):
        _same(w, g)
        assert all(np.isfinite(v).all() for v in g.values() if isinstance(v, np.ndarray))
