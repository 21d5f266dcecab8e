"""Host-side mirror of the reference's inputs / option handling (no GPU): option strings map to the
integers the solvers take in the order of the reference's option tables (reference
justdoit.py:5512-5534, 5655-5658), geometry set-up follows phase_angle() (justdoit.py:1453-1605),
and the minimal ATMSETUP reproduces the layer quantities the opacity stage consumes."""
import numpy as np
import pytest

from picaso_amd import disco
from picaso_amd import justdoit as jdi
from picaso_amd.atmsetup import ATMSETUP


def test_option_tables_match_reference_order():
    assert jdi.single_phase_options(False) == ["cahoy", "OTHG", "TTHG", "TTHG_ray"]
    assert jdi.multi_phase_options(False) == ["N=2", "N=1", "isotropic"]
    assert jdi.raman_options() == ["oklopcic", "pollack", "none"]
    assert jdi.toon_phase_coefficients(False) == ["quadrature", "eddington"]
    assert jdi.SH_scattering_options(False) == ["TTHG", "OTHG", "isotropic"]
    assert jdi.SH_psingle_form_options(False) == ["explicit", "legendre"]
    assert jdi.SH_rayleigh_options(False) == ["off", "on"]


def test_approx_maps_strings_to_solver_integers():
    c = jdi.inputs()
    c.approx(single_phase="OTHG", multi_phase="N=1", delta_eddington=False, raman="none", tthg_frac=[1, -1, 2],
             tthg_back=-0.5, tthg_forward=1, toon_coefficients="eddington", rt_method="SH", stream=4,
             w_single_form="OTHG", psingle_form="OTHG", w_multi_rayleigh="off", single_form="legendre",
             calculate_fluxes="on")
    a = c.inputs["approx"]
    toon, sh, com = a["rt_params"]["toon"], a["rt_params"]["SH"], a["rt_params"]["common"]
    assert (toon["single_phase"], toon["multi_phase"], toon["toon_coefficients"]) == (1, 1, 1)
    assert (sh["w_single_form"], sh["w_multi_form"], sh["psingle_form"]) == (1, 0, 1)
    assert (sh["w_single_rayleigh"], sh["w_multi_rayleigh"], sh["single_form"], sh["calculate_fluxes"]) == (1, 0, 1, 1)
    assert com["raman"] == 2 and com["stream"] == 4 and com["delta_eddington"] is False
    assert a["rt_method"] == "SH"
    with pytest.raises(Exception):
        c.approx(rt_method="disort")
    with pytest.raises(Exception):
        c.approx(rt_method="SH", stream=3)
    with pytest.raises(Exception):
        c.approx(tthg_frac=[1, -1])
    with pytest.raises(ValueError):
        c.approx(single_phase="nope")


def test_phase_angle_geometry():
    c = jdi.inputs()
    c.phase_angle(0, num_gangle=10)                      # halved by symmetry, snapped to 5..8
    g = c.inputs["disco"]
    assert (g["num_gangle"], g["num_tangle"]) == (5, 1)
    assert g["cos_theta"] == 1.0 and np.array_equal(g["ubar0"], g["ubar1"])
    gang, gw, tang, tw = disco.get_angles_1d(5)
    assert np.allclose(g["gweight"], gw) and np.isclose(np.sum(gw), 0.5, atol=0.01)
    c.phase_angle(np.pi / 3, num_gangle=6, num_tangle=4)
    g = c.inputs["disco"]
    assert g["ubar0"].shape == (6, 4) and np.isclose(g["cos_theta"], 0.5)
    for bad in (dict(phase=-0.1), dict(phase=7.0), dict(phase=0.5, num_tangle=1), dict(phase=0, num_gangle=1)):
        with pytest.raises(Exception):
            c.phase_angle(**bad)


def test_atmsetup_layer_quantities():
    c = jdi.inputs()
    c.gravity(gravity=2500.0)
    nlevel = 11
    p = np.logspace(-4, 1, nlevel)
    c.atmosphere(df={"pressure": p, "temperature": np.linspace(200, 900, nlevel), "H2": np.full(nlevel, 0.85),
                     "He": np.full(nlevel, 0.15)})
    atm = ATMSETUP(c.inputs)
    atm.planet.gravity, atm.planet.radius, atm.planet.mass = 2500.0, np.nan, np.nan
    atm.get_profile()
    atm.get_mmw()
    atm.get_altitude()
    atm.get_column_density()
    assert atm.c.nlevel == nlevel and atm.c.nlayer == nlevel - 1
    assert np.allclose(atm.layer["pressure"], np.sqrt(p[1:] * p[:-1]) * 1e6)      # bars -> dyn/cm2, log mean
    assert np.allclose(atm.layer["mmw"], 0.85 * 2.0156500642 + 0.15 * 4.0026032497, rtol=1e-14)   # main-isotope masses
    g_layer = np.full(nlevel - 1, 2500.0)
    g_layer[[0, -1]] = 1250.0                 # the reference's end layers (atmsetup.py:453)
    assert np.allclose(atm.layer["colden"], (p[1:] - p[:-1]) * 1e6 / g_layer)
    atm.get_clouds(np.linspace(1000, 2000, 7))
    assert atm.cloud_free and atm.layer["cloud"]["opd"].shape == (nlevel - 1, 7) and not atm.layer["cloud"]["opd"].any()


def test_spectrum_requires_inputs():
    c = jdi.inputs()
    with pytest.raises(Exception, match="atmosphere"):
        c.spectrum(None)
    c.atmosphere(df={"pressure": [1e-3, 1.0], "temperature": [300.0, 500.0]})
    with pytest.raises(Exception, match="gravity"):
        c.spectrum(None)
    with pytest.raises(Exception, match="dimension"):
        c.spectrum(None, dimension="2d")


def test_phase_angle_symmetry_quadrant():
    """symmetry=True keeps one quadrant with doubled weights (reference justdoit.py:1562-1602)."""
    c = jdi.inputs()
    c.phase_angle(0, num_gangle=6, num_tangle=4, symmetry=True)
    d = c.inputs["disco"]
    f = d["full_geometry"]
    assert (d["num_gangle"], d["num_tangle"], d["symmetry"]) == (3, 2, "true")
    assert d["ubar0"].shape == d["ubar1"].shape == (3, 2)
    assert np.array_equal(d["ubar0"], f["ubar0"][:3, :2]) and np.array_equal(d["gangle"], f["gangle"][:3])
    # the reference scales gweight by num_tangle/nt_uni and tweight by num_gangle/ng_uni (both 2 here)
    assert np.array_equal(d["gweight"], (4 / 2) * f["gweight"][:3])
    assert np.array_equal(d["tweight"], (6 / 3) * f["tweight"][:2])
    full = np.sum(np.outer(f["gweight"], f["tweight"]) * f["ubar0"])
    quad = np.sum(np.outer(d["gweight"], d["tweight"]) * d["ubar0"])
    assert np.isclose(full, quad, rtol=1e-14)
    c.phase_angle(0, num_gangle=6, num_tangle=4)
    assert c.inputs["disco"]["symmetry"] == "false" and "full_geometry" not in c.inputs["disco"]


@pytest.mark.parametrize("kw,msg", [(dict(phase=0.3, num_gangle=6, num_tangle=4), "non zero"),
                                    (dict(num_gangle=2, num_tangle=4), "num_gangle=2"),
                                    (dict(num_gangle=6, num_tangle=3), "num_tangle=3"),
                                    (dict(num_gangle=5, num_tangle=4), "num_gangle=5")])
def test_phase_angle_symmetry_errors(kw, msg):
    c = jdi.inputs()
    with pytest.raises(Exception, match=msg):
        c.phase_angle(symmetry=True, **kw)


def test_phase_curve_geometry():
    """Reflected light: the geometry of every phase; thermal: the phase-0 geometry for all phases
    (reference justdoit.py:1606-1660)."""
    from picaso_amd import disco
    phases = [0.0, 0.8, 2.1]
    c = jdi.inputs()
    c.phase_curve_geometry("reflected", phases, num_gangle=4, num_tangle=3)
    d = c.inputs["disco"]
    assert c.inputs["phase_angle"] == phases and d["calculation"] == "reflected"
    g, gw, t, tw = disco.get_angles_3d(4, 3)
    for p in phases:
        u0, u1, ct, lat, lon = disco.compute_disco(4, 3, g, t, p)
        assert np.array_equal(d[p]["ubar0"], u0) and np.array_equal(d[p]["ubar1"], u1) and d[p]["cos_theta"] == ct
        assert d[p]["symmetry"] == "false"
    c.phase_curve_geometry("thermal", phases, num_gangle=4, num_tangle=3)
    d = c.inputs["disco"]
    u0, u1, ct, lat, lon = disco.compute_disco(4, 3, g, t, 0.0)
    for p in phases:
        assert np.array_equal(d[p]["ubar0"], u0) and d[p]["cos_theta"] == ct
    with pytest.raises(Exception, match="thermal or reflected"):
        c.phase_curve_geometry("transmission", phases)
    with pytest.raises(Exception, match="greater than 2pi"):
        c.phase_curve_geometry("thermal", [0.0, 7.0])
    with pytest.raises(Exception, match="one profile per phase"):
        c.phase_curve(None)
