"""The reference's public Planck tables and its tidal-flux helpers (fluxes.blackbody, blackbody_integrated, chapman,
tidal_flux; reference picaso/fluxes.py:1609-1680, 3671-3751) against tests/golden/planck.npz -- outputs of the reference's
own functions on seeded inputs that include an overflowing exponential (tests/golden/make_golden.py:make_planck)."""
import collections
import os

import numpy as np
import pytest

from helpers import GOLDEN

Bundle = collections.namedtuple("InjectionBundle", "inject_beam beam_profile pm hratio wave_in")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "planck.npz"))


def _rel(a, b):
    """max relative error where the reference is non-zero; exact zeros (overflowed exponential) must be zeros"""
    z = b == 0.0
    assert np.array_equal(a[z], b[z])
    return float(np.max(np.abs(a[~z] - b[~z]) / np.abs(b[~z])))


def test_oracle_blackbody(gold, oracle):
    assert (gold["blackbody"] == 0.0).any(), "the fixture holds an overflowing exponential"
    assert _rel(oracle.blackbody(gold["t"], gold["w_cm"]), gold["blackbody"]) < 1e-14
    assert _rel(oracle.blackbody_integrated(gold["t"], gold["wno"], gold["dwno"]), gold["blackbody_integrated"]) < 1e-14


def test_chapman_and_tidal_flux_host(gold):
    from picaso_amd import fluxes
    p, cd, nlevel = gold["pressure"], gold["col_den"], gold["pressure"].size
    assert np.array_equal(np.array([fluxes.chapman(x, 0.01, 1.7) for x in p]), gold["chapman"])
    assert np.allclose(fluxes.chapman(p, 0.01, 1.7), gold["chapman"], rtol=1e-15, atol=0)
    a = fluxes.tidal_flux(450.0, nlevel, p, cd, Bundle(False, None, 0.01, 1.7, 2.5e6))
    assert np.array_equal(a, gold["tidal_chapman"])
    b = fluxes.tidal_flux(450.0, nlevel, p, cd, Bundle(True, gold["beam_profile"], 0.0, 0.0, 0.0))
    assert np.array_equal(b, gold["tidal_beam"])


@pytest.mark.gpu
def test_gpu_blackbody(gold):
    from picaso_amd import fluxes
    bb = fluxes.blackbody(gold["t"], gold["w_cm"])
    assert bb.shape == gold["blackbody"].shape
    assert _rel(bb, gold["blackbody"]) < 1e-13          # observed: see DESIGN section 9
    bi = fluxes.blackbody_integrated(gold["t"], gold["wno"], gold["dwno"])
    assert _rel(bi, gold["blackbody_integrated"]) < 1e-13
    # scalars as the reference's plotting code passes them (justplotit.py:1610)
    one = fluxes.blackbody(float(gold["t"][5]), float(gold["w_cm"][7]))
    assert one.shape == (1, 1) and one[0, 0] == bb[5, 7]


@pytest.mark.gpu
def test_gpu_blackbody_is_what_the_thermal_solver_uses(gold, oracle):
    """get_thermal_1d of an opaque isothermal non-scattering column returns 2 pi B at every angle... the table entry
    points and the in-sweep Planck evaluation are the same device functions: a large table against the oracle."""
    from picaso_amd import fluxes
    rng = np.random.default_rng(5)
    t = rng.uniform(60.0, 3000.0, 91)
    wno = np.linspace(40.0, 30000.0, 20011)
    bb = fluxes.blackbody(t, 1.0 / wno)
    assert _rel(bb, oracle.blackbody(t, 1.0 / wno)) < 1e-13
    dw = np.full(wno.size, wno[1] - wno[0])
    assert _rel(fluxes.blackbody_integrated(t, wno, dw), oracle.blackbody_integrated(t, wno, dw)) < 1e-13
