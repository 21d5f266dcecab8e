"""Disk geometry and Gauss-Chebyshev quadrature (counterpart of the reference ``picaso/disco.py``).

``get_angles_1d`` / ``get_angles_3d`` / ``compute_disco`` are tiny host-side table builders (numpy);
``compress_disco`` / ``compress_thermal`` run on the GPU through the C ABI.
"""
import ctypes

import numpy as np

from ._lib import check, context, f64, load, per_wave, ptr, serialized

# Half-sphere Gauss abscissae/weights, Abramowitz & Stegun Table 25.8 (k = 0), n = 5..8
# (the same table the reference hard-codes, disco.py:67-84).
_AS_25_8 = {
    5: ((0.0985350858, 0.3045357266, 0.5620251898, 0.8019865821, 0.9601901429),
        (0.0157479145, 0.0739088701, 0.1463869871, 0.1671746381, 0.0967815902)),
    6: ((0.0730543287, 0.2307661380, 0.4413284812, 0.6630153097, 0.8519214003, 0.9706835728),
        (0.0087383018, 0.0439551656, 0.0986611509, 0.1407925538, 0.1355424972, 0.0723103307)),
    7: ((0.0562625605, 0.1802406917, 0.3526247171, 0.5471536263, 0.7342101772, 0.8853209468,
         0.9775206136),
        (0.0052143622, 0.0274083567, 0.0663846965, 0.1071250657, 0.1273908973, 0.1105092582,
         0.0559673634)),
    8: ((0.0446339553, 0.1443662570, 0.2868247571, 0.4548133152, 0.6280678354, 0.7856915206,
         0.9086763921, 0.9822200849),
        (0.0032951914, 0.0178429027, 0.0454393195, 0.0791995995, 0.1060473594, 0.1125057995,
         0.0911190236, 0.0445508044)),
}


def get_angles_1d(ngauss):
    """Half-sphere Gauss angles for the symmetric (zero phase) 1-D geometry
    (reference disco.py:52-89).  Returns gangle, gweight, tangle, tweight."""
    if ngauss not in _AS_25_8:
        raise Exception("Please enter ngauss=5,6,7 or 8.")
    g, w = _AS_25_8[ngauss]
    return np.array(g), np.array(w), np.array([0]), np.array([1])


def get_angles_3d(num_gangle, num_tangle):
    """Gauss-Legendre x Chebyshev angles for the full disk (reference disco.py:92-115)."""
    i = np.linspace(1, num_tangle, num_tangle)
    tangle = np.cos(i * np.pi / (num_tangle + 1))
    tweight = np.pi / (num_tangle + 1) * np.sin(i * np.pi / (num_tangle + 1)) ** 2.0
    gangle, gweight = np.polynomial.legendre.leggauss(num_gangle)
    return gangle, gweight, tangle, tweight


def compute_disco(ng, nt, gangle, tangle, phase_angle):
    """Incident / outgoing cosines per facet (reference disco.py:7-50).
    Returns ubar0, ubar1, cos_theta, latitude, longitude."""
    cos_theta = np.cos(phase_angle)
    arg = (gangle - (cos_theta - 1.0) / (cos_theta + 1.0)) / (2.0 / (cos_theta + 1))
    longitude = np.arcsin(arg) if phase_angle <= np.pi else -np.arcsin(arg)
    colatitude = np.arccos(tangle)
    latitude = np.pi / 2 - colatitude
    f = np.sin(colatitude)
    ubar0 = np.outer(np.cos(longitude - phase_angle), f)
    ubar1 = np.outer(np.cos(longitude), f)
    return ubar0, ubar1, cos_theta, latitude, longitude


@serialized
def compress_disco(nwno, cos_theta, xint_at_top, gweight, tweight, F0PI):
    """Disk-integrated albedo (reference disco.py:117-149)."""
    ctx = context()
    gw, tw = f64(gweight), f64(tweight)
    x = f64(xint_at_top, (len(gw), len(tw), nwno))
    f0 = per_wave(F0PI, nwno)
    out = np.zeros(nwno)
    check(load().picaso_compress_disco(ctx, ctypes.c_int(nwno), ctypes.c_double(cos_theta), ptr(x),
                                       ptr(gw), ctypes.c_int(len(gw)), ptr(tw),
                                       ctypes.c_int(len(tw)), ptr(f0), ptr(out)), ctx)
    return out


@serialized
def compress_thermal(nwno, flux_at_top, gweight, tweight):
    """Disk-integrated thermal flux; 3-D ``(ng,nt,nwno)`` or 4-D ``(ng,nt,nlevel,nwno)`` input
    (reference disco.py:151-181)."""
    ctx = context()
    gw, tw = f64(gweight), f64(tweight)
    x = f64(flux_at_top)
    inner = x.shape[2:]
    out = np.zeros(inner)
    check(load().picaso_compress_thermal(ctx, ctypes.c_size_t(int(np.prod(inner))), ptr(x), ptr(gw),
                                         ctypes.c_int(len(gw)), ptr(tw), ctypes.c_int(len(tw)),
                                         ptr(out)), ctx)
    return out
