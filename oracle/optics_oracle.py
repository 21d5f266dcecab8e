"""numpy restatement of the reference's opacity pre-stage -- TEST INFRASTRUCTURE.

Restates, from plain arrays (no sqlite, no classes):
  * interp_molecular : RetrieveOpacities.get_opacities / get_opacities_nearest arithmetic
                       (reference picaso/optics.py:2277-2294, :2350-2351)
  * compute_opacity  : reference picaso/optics.py:26-431 (TAUGAS/TAURAY sums :144-277, Raman
                       clip :294, mixing :327-354, test_mode :372-399, delta-Eddington :401-431)
Pinned against tests/golden/optics.npz (outputs of the reference's own source on the synthetic DB
tests/golden/synthetic_opacities.db) by tests/test_oracle_golden.py.
"""
import numpy as np

AVOGADRO = 6.02214086e+23


def interp_molecular(rows4, t_interp, p_interp, i_ll, i_hl, i_hh, i_lh):
    """rows4: (npt, nwno) raw kappa table addressed by 0-based row; returns (nlayer, nwno)."""
    nlayer = len(t_interp)
    out = np.zeros((nlayer, rows4.shape[1]))
    lg = np.log10(np.where(rows4 != 0, rows4, 1e-50))
    for ind in range(nlayer):
        t, p = t_interp[ind], p_interp[ind]
        cx = 10 ** (((1 - t) * (1 - p) * lg[i_ll[ind]]) + ((t) * (1 - p) * lg[i_hl[ind]]) +
                    ((t) * (p) * lg[i_hh[ind]]) + ((1 - t) * (p) * lg[i_lh[ind]]))
        out[ind] = cx * AVOGADRO
    return out


def compute_opacity(taugas, tauray, taucld, w0_cld, g0_cld, raman_factor, stream=2,
                    delta_eddington=True, test_mode=None):
    """Mixing half of compute_opacity from the three optical-depth components."""
    nlayer, nwno = taugas.shape
    DTAU = taugas + tauray + taucld
    with np.errstate(invalid="ignore", divide="ignore"):
        ftau_cld = (w0_cld * taucld) / (w0_cld * taucld + tauray)
        ftau_ray = tauray / (tauray + w0_cld * taucld)
    COSB = g0_cld
    GCOS2 = 0.5 * ftau_ray
    W0 = (tauray * raman_factor + taucld * w0_cld) / (taugas + tauray + taucld)
    W0_no_raman = (tauray * 0.99999 + taucld * w0_cld) / (taugas + tauray + taucld)
    TAU = np.zeros((nlayer + 1, nwno))
    TAU[1:] = np.cumsum(DTAU, axis=0)
    if test_mode is not None:
        if test_mode == "rayleigh":
            DTAU = tauray.copy()
            GCOS2 = np.zeros(DTAU.shape) + 0.5
            ftau_ray = np.zeros(DTAU.shape) + 1.0
            ftau_cld = np.zeros(DTAU.shape)
        else:
            DTAU = np.zeros(DTAU.shape) + taucld
            GCOS2 = np.zeros(DTAU.shape)
            ftau_ray = np.zeros(DTAU.shape)
            ftau_cld = np.zeros(DTAU.shape) + 1.
        w0c = np.where(w0_cld <= 0, 1e-10, w0_cld)
        DTAU = np.where(DTAU <= 0, 1e-10, DTAU)
        COSB = g0_cld + 0 * DTAU
        W0 = w0c + 0 * DTAU
        W0_no_raman = W0
        TAU = np.zeros((nlayer + 1, nwno))
        TAU[1:] = np.cumsum(DTAU, axis=0)
    if delta_eddington:
        f_deltaM = COSB ** stream
        w0_dedd = W0 * (1. - f_deltaM) / (1.0 - W0 * f_deltaM)
        cosb_dedd = (COSB - f_deltaM) / (1. - f_deltaM)
        dtau_dedd = DTAU * (1. - W0 * f_deltaM)
        tau_dedd = np.zeros((nlayer + 1, nwno))
        tau_dedd[1:] = np.cumsum(dtau_dedd, axis=0)
        return (dtau_dedd, tau_dedd, w0_dedd, cosb_dedd, ftau_cld, ftau_ray, GCOS2, DTAU, TAU, W0,
                COSB, W0_no_raman, f_deltaM)
    return (DTAU, TAU, W0, COSB, ftau_cld, ftau_ray, GCOS2, DTAU, TAU, W0, COSB, W0_no_raman,
            0 * COSB)
