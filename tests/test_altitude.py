"""Host-side hydrostatic altitude / column density against vectors produced by the reference's own
``ATMSETUP.get_altitude`` / ``get_column_density`` (tests/golden/make_golden.py altitude)."""
import os

import numpy as np
import pytest

from picaso_amd.atmsetup import ATMSETUP, molecular_weight

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "altitude.npz"))
CASES = sorted({k.split("/")[0] for k in G.files} - {"weights"})


@pytest.mark.parametrize("case", CASES)
def test_altitude_matches_reference(case):
    g = {k.split("/", 1)[1]: G[k] for k in G.files if k.startswith(case + "/")}
    atm = ATMSETUP({})
    atm.planet.radius, atm.planet.mass = float(g["radius"]), float(g["mass"])
    atm.planet.gravity = float(g["gravity"])
    atm.level.update(mmw=g["mmw"], temperature=g["temperature"], pressure=g["pressure"])
    atm.c.nlevel, atm.c.nlayer = len(g["mmw"]), len(g["mmw"]) - 1
    atm.get_altitude(p_reference=float(g["p_reference"]))
    atm.get_column_density()
    for key, got in (("z", atm.level["z"]), ("dz", atm.level["dz"]),
                     ("scale_height", atm.level["scale_height"]),
                     ("layer_gravity", atm.layer["gravity"]), ("colden", atm.layer["colden"])):
        np.testing.assert_allclose(got, g[key], rtol=1e-14, atol=0, equal_nan=True, err_msg=key)


def test_end_layers_carry_half_gravity():
    # the reference averages the level gravity before filling its two end levels (atmsetup.py:453)
    g = {k.split("/", 1)[1]: G[k] for k in G.files if k.startswith("const_g/")}
    lg = g["layer_gravity"]
    assert lg[0] == 0.5 * g["gravity"] and lg[-1] == 0.5 * g["gravity"]
    assert np.all(lg[1:-1] == g["gravity"])


def test_molecular_weights_match_reference():
    """Main-isotope masses (reference ATMSETUP.get_weights, atmsetup.py:285-342), not standard
    atomic weights: H2 = 2.01565."""
    for name, want in zip(G["weights/names"], G["weights/values"]):
        if want == 0.0:                       # no element token: the reference carries it at weight 0
            with pytest.raises(KeyError):
                molecular_weight(str(name))
        else:
            assert molecular_weight(str(name)) == pytest.approx(float(want), rel=1e-15), name
    assert molecular_weight("H2") == pytest.approx(2.0156500642, rel=1e-15)


def test_constant_gravity_with_radius_matches_level_loop():
    """The element-wise constant-gravity path against the level-by-level recurrence it replaces
    (z stays finite when a radius is given and constant_gravity=True is asked for)."""
    rng = np.random.default_rng(0)
    n = 23
    p = np.logspace(-5, 2, n) * 1e6
    t = 200.0 + 900.0 * rng.random(n)
    w = 2.2 + 0.2 * rng.random(n)
    g, k_b, amu, r0 = 1234.0, 1.380649e-16, 1.66053906660e-24, 7.0e9
    for pref_bar in (1e-9, 1.0, 1e4):
        atm = ATMSETUP({})
        atm.planet.radius, atm.planet.mass, atm.planet.gravity = r0, 1.9e30, g
        atm.level.update(mmw=w, temperature=t, pressure=p)
        atm.c.nlevel, atm.c.nlayer = n, n - 1
        atm.get_altitude(p_reference=pref_bar, constant_gravity=True)
        pref = min(pref_bar * 1e6, p.max())
        iref = int(np.argmax(p >= pref))
        z, dz = np.zeros(n) + r0, np.zeros(n)
        for i in range(iref, n - 1):
            dz[i] = k_b * t[i] / (w[i] * amu * g) * np.log(p[i + 1] / p[i])
            z[i + 1] = z[i] - dz[i]
        for i in range(iref, 0, -1):
            dz[i] = k_b * t[i] / (w[i] * amu * g) * np.log(p[i] / p[i - 1])
            z[i - 1] = z[i] + dz[i]
        dz[0], dz[-1] = dz[1], dz[-2]
        assert np.array_equal(atm.level["z"], z) and np.array_equal(atm.level["dz"], dz), pref_bar
        assert np.all(np.diff(atm.level["z"]) < 0)
