"""One spectrum on several GPUs from the product API: ``inputs.spectrum(opa, ..., devices=N)`` /
``picaso(..., devices=N)`` / ``phase_curve(..., devices=N)`` (reference fan-out: justdoit.py:4741-4777).

A 1-GPU box exercises the whole sharded path by listing device 0 several times (``devices=[0, 0, 0]``: one
context -- one stream, its own opacity-table block -- per entry), ragged blocks included: every function on the
path is pointwise in wavelength, so the gathered result must equal the unsharded one bit for bit.  The RCCL
gather (``gather='rccl'``, ``picaso_comm_init_all`` + grouped all-gather) takes one rank per device and runs
here with one device; with two or more GPUs visible the multi-device tests run as well.
"""
import os
import sqlite3

import numpy as np
import pytest

from helpers import GOLDEN

DB = os.path.join(GOLDEN, "synthetic_opacities.db")


@pytest.fixture(scope="module")
def og():
    return np.load(os.path.join(GOLDEN, "optics.npz"))


def _case(og, jdi, cloud=True, star=True):
    case = jdi.inputs()
    case.phase_angle(0)
    case.gravity(gravity=float(og["in/gravity"]), radius=7.1e9, mass=1.9e30)
    prof = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"]}
    for k in ("H2", "He", "H2O", "CH4"):
        prof[k] = og["in/mix/" + k]
    case.atmosphere(df=prof)
    if cloud:
        case.clouds(df={"opd": og["in/cld_opd"], "w0": og["in/cld_w0"], "g0": og["in/cld_g0"]})
    nwno = len(og["in/wno"])
    if star:
        case.star(relative_flux=1.0 + 0.3 * np.sin(np.arange(nwno) / 7.0), radius=6.9e10, semi_major=7.5e12)
    case.surface_reflect(0.1 + 0.2 * np.cos(np.arange(nwno) / 11.0) ** 2)
    case.approx(raman="none", delta_eddington=True)
    return case


def _same(a, b, path=""):
    if isinstance(a, dict):
        assert isinstance(b, dict) and list(a.keys()) == list(b.keys()), path
        for k in a:
            _same(a[k], b[k], path + "/" + str(k))
    elif isinstance(a, np.ndarray):
        assert isinstance(b, np.ndarray) and a.shape == b.shape, path
        assert np.array_equal(a, b, equal_nan=True), path
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, path + "[%d]" % i)
    else:
        assert (a == b) or (a != a and b != b), (path, a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [1, [0], [0, 0], [0, 0, 0]])
@pytest.mark.parametrize("calc", ["reflected+thermal", "reflected", "thermal+transmission"])
def test_devices_equals_single_device_bit_for_bit(og, devices, calc):
    from picaso_amd import justdoit as jdi
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    want = _case(og, jdi).spectrum(opa, calculation=calc, full_output=True)
    got = _case(og, jdi).spectrum(opa, calculation=calc, full_output=True, devices=devices)
    _same(want, got)
    # the blocks really were separate opacity objects on their own contexts
    n = devices if isinstance(devices, int) else len(devices)
    shards = opa._shards[tuple(range(n)) if isinstance(devices, int) else tuple(devices)]
    assert [hi - lo for lo, hi, _ in shards] == [len(x) for x in np.array_split(np.arange(opa.nwno), n)]
    if n > 1:
        assert len({s.ctx.value for _, _, s in shards}) == n and all(s.nwno < opa.nwno for _, _, s in shards)


@pytest.mark.gpu
def test_devices_gather_rccl_one_device(og):
    """``gather='rccl'`` through picaso_comm_init_all + picaso_all_gather_group_dev (one rank here)."""
    from picaso_amd import justdoit as jdi
    opa = jdi.opannection(filename_db=DB, query_method="nearest")
    want = _case(og, jdi).spectrum(opa, calculation="reflected+thermal")
    got = _case(og, jdi).spectrum(opa, calculation="reflected+thermal", devices=[0], gather="rccl")
    _same(want, got)


@pytest.mark.gpu
def test_device_group_collectives_one_device():
    """The grouped (one thread, all communicators) collectives with the one communicator a 1-GPU box has."""
    from picaso_amd import _lib, device, sharding
    grp = sharding.device_group([0])
    ctx = _lib.context(0)
    loc = device.DeviceArray.from_host(np.arange(37.0), ctx)
    full = device.DeviceArray.zeros((37,), ctx)
    grp.all_gather_spectrum([loc], [full], 37)
    device.sync(ctx)
    assert np.array_equal(full.to_host(), np.arange(37.0))
    assert grp.max([3.5]) == [3.5]
    grp.barrier()
    with pytest.raises(_lib.PicasoHipError):
        sharding.DeviceGroup([0, 0])


@pytest.mark.gpu
def test_devices_more_than_visible_is_an_error(og):
    from picaso_amd import _lib
    from picaso_amd import justdoit as jdi
    opa = jdi.opannection(filename_db=DB)
    n = _lib.device_count()
    with pytest.raises(_lib.PicasoHipError, match="needs %d GPU" % (n + 1)):
        _case(og, jdi).spectrum(opa, devices=n + 1)


@pytest.mark.gpu
@pytest.mark.parametrize("fly", [False, True])
def test_devices_correlated_k_and_patchy_clouds(og, fly):
    """A correlated-k opacity object (Gauss index fastest inside a wavelength; premixed table and per-gas tables
    mixed on the fly) cut into three ragged blocks, with a patchy cloud deck."""
    import test_ck_optics as tck
    from picaso_amd import justdoit as jdi
    ck = np.load(os.path.join(GOLDEN, "ck.npz"))
    opa = tck._ck_class(ck, fly=fly)

    def run(devices):
        case = _case(og, jdi, star=False)
        case.inputs.pop("surface_reflect", None)
        case.inputs.pop("hard_surface", None)
        case.clouds(df={"opd": og["in/cld_opd"], "w0": og["in/cld_w0"], "g0": og["in/cld_g0"]}, do_holes=True,
                    fhole=0.3, fthin_cld=0.2)
        return case.spectrum(opa, calculation="reflected+thermal", devices=devices)
    _same(run(None), run([0, 0, 0]))


@pytest.mark.gpu
def test_phase_curve_devices_round_robin(og):
    from picaso_amd import justdoit as jdi
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    ng, nt = 3, 2
    phases = [0.0, 0.9, 2.0, 2.6]

    def profile(k):
        dT = 40.0 * k * np.cos(np.arange(ng))[None, :, None] * np.ones((1, 1, nt))
        pr = {"pressure": og["in/plevel_bar"], "temperature": og["in/tlevel"][:, None, None] + dT}
        for m in ("H2", "He", "H2O", "CH4"):
            pr[m] = og["in/mix/" + m]
        return pr

    def curve(devices):
        case = jdi.inputs()
        case.gravity(gravity=float(og["in/gravity"]))
        case.approx(raman="none")
        case.phase_curve_geometry("reflected", phases, num_gangle=ng, num_tangle=nt)
        case.atmosphere_4d([profile(k) for k in range(len(phases))])
        return case.phase_curve(opa, devices=devices)
    a, b = curve(None), curve([0, 0])
    assert list(a.keys()) == list(b.keys()) == phases
    for ph in phases:
        _same(a[ph], b[ph])
    assert len(opa._replicas[(0, 0)]) == 2 and opa._replicas[(0, 0)][0] is opa


@pytest.mark.gpu
def test_devices_3d_spectrum(og):
    from picaso_amd import justdoit as jdi
    ng, nt = 2, 3
    opa = jdi.opannection(filename_db=DB, query_method="linear")

    def run(devices):
        case = jdi.inputs()
        case.phase_angle(np.pi / 3, num_gangle=ng, num_tangle=nt)
        case.gravity(gravity=float(og["in/gravity"]))
        prof = {"pressure": og["in/plevel_bar"],
                "temperature": np.repeat(og["in/tlevel"][:, None, None], ng, 1).repeat(nt, 2)}
        for k in ("H2", "He", "H2O", "CH4"):
            prof[k] = og["in/mix/" + k]
        case.atmosphere_3d(prof)
        cld = {k: np.repeat(og["in/cld_" + k][:, :, None, None], ng, 2).repeat(nt, 3) for k in ("opd", "w0", "g0")}
        case.clouds_3d(cld)
        case.approx(raman="none")
        return case.spectrum(opa, calculation="reflected+thermal", dimension="3d", full_output=True, devices=devices)
    _same(run(None), run([0, 0]))


@pytest.mark.gpu
def test_devices_two_real_gpus(og):
    """With two GPUs visible: blocks on two devices, host gather and RCCL gather, bit-identical to one GPU."""
    from picaso_amd import _lib
    from picaso_amd import justdoit as jdi
    if _lib.device_count() < 2:
        pytest.skip("one GPU visible")
    opa = jdi.opannection(filename_db=DB, query_method="linear")
    want = _case(og, jdi).spectrum(opa, calculation="reflected+thermal", full_output=True)
    _same(want, _case(og, jdi).spectrum(opa, calculation="reflected+thermal", full_output=True, devices=2))
    _same(want, _case(og, jdi).spectrum(opa, calculation="reflected+thermal", full_output=True, devices=2,
                                        gather="rccl"))
