"""numpy restatement of the reference's opacity pre-stage -- TEST INFRASTRUCTURE.

Restates, from plain arrays (no sqlite, no classes):
  * interp_molecular : RetrieveOpacities.get_opacities / get_opacities_nearest arithmetic
                       (reference picaso/optics.py:2277-2294, :2350-2351)
  * compute_opacity  : reference picaso/optics.py:26-431 (TAUGAS/TAURAY sums :144-277, Raman
                       clip :294, mixing :327-354, test_mode :372-399, delta-Eddington :401-431)
  * pre_mix_ck       : RetrieveCKs.get_pre_mix_ck (optics.py:1081-1161), premixed correlated-k
  * continuum_ck     : RetrieveCKs.get_continuum interpolation (optics.py:1411-1428, 1470-1491)
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
    nlayer = taugas.shape[0]          # trailing axes: (nwno,) or (nwno, ngauss)
    lvl_shape = (nlayer + 1,) + tuple(np.broadcast_shapes(taugas.shape, tauray.shape)[1:])
    DTAU = taugas + tauray + taucld
    with np.errstate(invalid="ignore", divide="ignore"):
        ftau_cld = (w0_cld * taucld) / (w0_cld * taucld + tauray)
        ftau_ray = tauray / (tauray + w0_cld * taucld)
    COSB = g0_cld
    GCOS2 = 0.5 * ftau_ray
    W0 = (tauray * raman_factor + taucld * w0_cld) / (taugas + tauray + taucld)
    W0_no_raman = (tauray * 0.99999 + taucld * w0_cld) / (taugas + tauray + taucld)
    TAU = np.zeros(lvl_shape)
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
        TAU = np.zeros(lvl_shape)
        TAU[1:] = np.cumsum(DTAU, axis=0)
    if delta_eddington:
        f_deltaM = COSB ** stream
        w0_dedd = W0 * (1. - f_deltaM) / (1.0 - W0 * f_deltaM)
        cosb_dedd = (COSB - f_deltaM) / (1. - f_deltaM)
        dtau_dedd = DTAU * (1. - W0 * f_deltaM)
        tau_dedd = np.zeros(lvl_shape)
        tau_dedd[1:] = np.cumsum(dtau_dedd, axis=0)
        return (dtau_dedd, tau_dedd, w0_dedd, cosb_dedd, ftau_cld, ftau_ray, GCOS2, DTAU, TAU, W0,
                COSB, W0_no_raman, f_deltaM)
    return (DTAU, TAU, W0, COSB, ftau_cld, ftau_ray, GCOS2, DTAU, TAU, W0, COSB, W0_no_raman,
            0 * COSB)


def pre_mix_ck_indices(player_bar, tlayer, press_grid, temp_grid, nc_p):
    """Bracketing indices / weights of RetrieveCKs.get_pre_mix_ck (reference optics.py:1087-1150):
    ``press_grid`` / ``temp_grid`` are the unique grid values (ascending), ``nc_p[it]`` the number of
    pressures available at temperature ``it`` (ragged grid)."""
    t_inv = 1 / np.asarray(tlayer, dtype=float)
    p_log = np.log10(np.asarray(player_bar, dtype=float))
    p_log_grid = np.log10(press_grid[press_grid > 0])
    t_inv_grid = 1 / np.asarray(temp_grid, dtype=float)
    t_low_ind = []
    for i in t_inv:
        find = np.where(t_inv_grid > i)[0]
        t_low_ind += [0] if len(find) == 0 else [find[-1]]
    t_low_ind = np.array(t_low_ind)
    t_low_ind[t_low_ind == (len(t_inv_grid) - 1)] = len(t_inv_grid) - 2
    t_hi_ind = t_low_ind + 1
    p_low_ind = []
    for i in p_log:
        find = np.where(p_log_grid <= i)[0]
        p_low_ind += [0] if len(find) == 0 else [find[-1]]
    p_low_ind = np.array(p_low_ind)
    for i in range(len(p_low_ind)):
        p_low_ind[i] = min(p_low_ind[i], nc_p[t_hi_ind[i]] - 3)
    p_hi_ind = p_low_ind + 1
    t_interp = (t_inv - t_inv_grid[t_low_ind]) / (t_inv_grid[t_hi_ind] - t_inv_grid[t_low_ind])
    p_interp = (p_log - p_log_grid[p_low_ind]) / (p_log_grid[p_hi_ind] - p_log_grid[p_low_ind])
    return t_interp, p_interp, p_low_ind, p_hi_ind, t_low_ind, t_hi_ind


def pre_mix_ck(player_bar, tlayer, press_grid, temp_grid, nc_p, ln_kappa):
    """molecular_opa (nlayer, nwno, ngauss) from the ln(kappa)[p, t, wno, gauss] table
    (reference optics.py:1152-1159)."""
    t, p, pl, ph, tl, th = pre_mix_ck_indices(player_bar, tlayer, press_grid, temp_grid, nc_p)
    t = t[:, None, None]
    p = p[:, None, None]
    k = np.exp(((1 - t) * (1 - p) * ln_kappa[pl, tl, :, :]) + ((t) * (1 - p) * ln_kappa[pl, th, :, :]) +
               ((t) * (p) * ln_kappa[ph, th, :, :]) + ((1 - t) * (p) * ln_kappa[ph, tl, :, :]))
    return k * AVOGADRO


def continuum_ck_indices(tlayer, cia_temps):
    """Bracketing CIA temperatures of RetrieveCKs.get_continuum (reference optics.py:1411-1428) as
    indices into the sorted temperature list, and the 1/T interpolation weight (:1474-1478)."""
    st = np.sort(np.asarray(cia_temps, dtype=float))
    lo = np.zeros(len(tlayer), dtype=int)
    for i, t in enumerate(tlayer):
        if t <= st[0]:
            lo[i] = 0
        elif t >= st[-1]:
            lo[i] = len(st) - 2
        else:
            lo[i] = np.where(st - t <= 0)[0][-1]
    hi = lo + 1
    t_inv = 1 / np.asarray(tlayer, dtype=float)
    t_interp = (t_inv - 1 / st[lo]) / (1 / st[hi] - 1 / st[lo])
    return lo, hi, t_interp


def continuum_ck(tlayer, cia_temps, table):
    """continuum_opa (nlayer, nwno): exp((1-t) ln k_lo + t ln k_hi) (reference optics.py:1486-1489);
    ``table`` is (n_cia_temps, nwno) in sorted-temperature order."""
    lo, hi, t = continuum_ck_indices(tlayer, cia_temps)
    t = t[:, None]
    return np.exp((1 - t) * np.log(table[lo]) + t * np.log(table[hi]))
