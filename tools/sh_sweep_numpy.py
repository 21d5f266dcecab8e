"""numpy statement of the single top-down BLOCK sweep for the SH4 / SH2 solvers (any float dtype).

The reference assembles an 11-diagonal (4 unknowns per layer) system per (wavelength, angle) and
solves it with LAPACK dgbsv.  Per layer the unknowns split into decaying-mode coefficients
d = (X0, X2) and growing-mode coefficients u = (X1, X3) = E v, E = diag(exp(-lam1 dtau),
exp(-lam2 dtau)); with Mn = [[p1mn,p2mn],[q1mn,q2mn]], Pl = [[p1pl,p2pl],[q1pl,q2pl]] the four
moment fluxes at the top / bottom of a layer are
    top :  Fmn = Mn d + Pl E v + zmn_dn      Fpl = Pl d + Mn E v + zpl_dn
    bot :  Fmn = Mn E d + Pl v + zmn_up      Fpl = Pl E d + Mn v + zpl_up
(reference setup_4_stream_fluxes rows, fluxes.py:3469-3543).  Exactly as in the Toon kernels we
carry the relation d_i = delta_i - R_i v_i (R: 2x2) downwards together with the TOA functional
J = kappa + zeta . v_i (reference source-function integrals fluxes.py:2898-2970), so the banded
matrix is never formed and no pivoting across layers is needed (the 2x2 blocks are the physical
reflection operators; all exponentials that appear are decaying).  Development / test aid for
picaso_amd/csrc/sh.hip; checked by tests/test_single_sweep_numpy.py.
"""
import numpy as np

PI = np.pi


def _inv2(M):
    det = M[..., 0, 0] * M[..., 1, 1] - M[..., 0, 1] * M[..., 1, 0]
    out = np.empty_like(M)
    out[..., 0, 0] = M[..., 1, 1] / det
    out[..., 1, 1] = M[..., 0, 0] / det
    out[..., 0, 1] = -M[..., 0, 1] / det
    out[..., 1, 0] = -M[..., 1, 0] / det
    return out


def _mm(A, B):
    return np.einsum("...ij,...jk->...ik", A, B)


def _mv(A, x):
    return np.einsum("...ij,...j->...i", A, x)


def _clip(x):
    return np.clip(x, -35.0, 35.0)


def legP(mu):
    return np.array([1, mu, (3 * mu ** 2 - 1) / 2, (5 * mu ** 3 - 3 * mu) / 2])


def sh4_layer(a, dtau):
    """Per-layer mode quantities from a_l (4, nwno): lam1, lam2, Mn, Pl, E, Amat (4x4 columns m)."""
    a0, a1, a2, a3 = a
    beta = a0 * a1 + 4 * a0 * a3 / 9 + a2 * a3 / 9
    gama = a0 * a1 * a2 * a3 / 9
    disc = np.sqrt(beta ** 2 - 4 * gama)
    lam1 = np.sqrt((beta + disc) / 2)
    lam2 = np.sqrt((beta - disc) / 2)
    R1, R2 = -a0 / lam1, -a0 / lam2
    Q1, Q2 = 0.5 * (a0 * a1 / lam1 ** 2 - 1), 0.5 * (a0 * a1 / lam2 ** 2 - 1)
    S1, S2 = -3 / (2 * a3) * (a0 * a1 / lam1 - lam1), -3 / (2 * a3) * (a0 * a1 / lam2 - lam2)
    tp = 2 * PI
    p1pl, p2pl = (0.5 + R1 + 5 * Q1 / 8) * tp, (0.5 + R2 + 5 * Q2 / 8) * tp
    q1pl, q2pl = (-0.125 + 5 * Q1 / 8 + S1) * tp, (-0.125 + 5 * Q2 / 8 + S2) * tp
    p1mn, p2mn = (0.5 - R1 + 5 * Q1 / 8) * tp, (0.5 - R2 + 5 * Q2 / 8) * tp
    q1mn, q2mn = (-0.125 + 5 * Q1 / 8 - S1) * tp, (-0.125 + 5 * Q2 / 8 - S2) * tp
    nw = a0.shape[0]
    Mn = np.empty((nw, 2, 2), dtype=a0.dtype)
    Pl = np.empty((nw, 2, 2), dtype=a0.dtype)
    Mn[:, 0, 0], Mn[:, 0, 1], Mn[:, 1, 0], Mn[:, 1, 1] = p1mn, p2mn, q1mn, q2mn
    Pl[:, 0, 0], Pl[:, 0, 1], Pl[:, 1, 0], Pl[:, 1, 1] = p1pl, p2pl, q1pl, q2pl
    E = np.stack([np.exp(-_clip(lam1 * dtau)), np.exp(-_clip(lam2 * dtau))], axis=-1)   # (nw, 2)
    one = np.ones_like(a0)
    # A[j][m] columns: m = 0 (d0), 1 (u0), 2 (d1), 3 (u1)
    Amat = np.array([[one, one, one, one], [R1, -R1, R2, -R2], [Q1, Q1, Q2, Q2], [S1, -S1, S2, -S2]])
    return lam1, lam2, Mn, Pl, E, Amat, beta, gama


def reflected_sh4(nlevel, nwno, dtau, tau, w0, ftau_cld, ftau_ray, f_deltaM, dtau_og, tau_og, w0_og,
                  cosb_og, rs, u0a, u1a, ct, F, w_single_form, w_multi_form, psingle_form,
                  w_single_rayleigh, w_multi_rayleigh, psingle_rayleigh, fa, fb, fc_, cb_, cf_,
                  b_top, single_form, compound=True):
    n = nlevel - 1
    stream = 4
    out = np.zeros((len(u0a), nwno), dtype=dtau.dtype)
    fd_run = f_deltaM.copy()
    for k, (u0, u1) in enumerate(zip(u0a, u1a)):
        Pu0, Pu1 = legP(-u0), legP(u1)
        # ---- Legendre weights (fluxes.py:2803-2840) ----
        wsg = np.ones((4, n, nwno), dtype=dtau.dtype)
        wmu = np.ones((4, n, nwno), dtype=dtau.dtype)
        if w_single_form == 1 or w_multi_form == 1:
            for l in range(1, 4):
                w = (2 * l + 1) * cosb_og ** l
                if w_single_form == 1:
                    wsg[l] = (w - (2 * l + 1) * fd_run) / (1 - fd_run)
                if w_multi_form == 1:
                    wmu[l] = (w - (2 * l + 1) * fd_run) / (1 - fd_run)
        if w_single_form == 0 or w_multi_form == 0:
            gf, gb = cf_ * cosb_og, cb_ * cosb_og
            f = fa + fb * gb ** fc_
            fac = (f * cf_ ** stream + (1 - f) * cb_ ** stream)
            fd_run = fd_run * fac if compound else f_deltaM * fac
            for l in range(1, 4):
                w = (2 * l + 1) * (f * gf ** l + (1 - f) * gb ** l)
                if w_single_form == 0:
                    wsg[l] = (w - (2 * l + 1) * fd_run) / (1 - fd_run)
                if w_multi_form == 0:
                    wmu[l] = (w - (2 * l + 1) * fd_run) / (1 - fd_run)
        if w_single_rayleigh == 1:
            wsg[1:] = wsg[1:] * ftau_cld
            wsg[2] = wsg[2] + 0.5 * ftau_ray
        if w_multi_rayleigh == 1:
            wmu[1:] = wmu[1:] * ftau_cld
            wmu[2] = wmu[2] + 0.5 * ftau_ray
        if single_form == 0:
            if psingle_form == 1:
                p = (1 - cosb_og ** 2) / (np.sqrt(1 + cosb_og ** 2 + 2 * cosb_og * ct) ** 3)
            else:
                gf, gb = cf_ * cosb_og, cb_ * cosb_og
                f = fa + fb * gb ** fc_
                p = (f * (1 - gf ** 2) / np.sqrt((1 + gf ** 2 + 2 * gf * ct) ** 3)
                     + (1 - f) * (1 - gb ** 2) / np.sqrt((1 + gb ** 2 + 2 * gb * ct) ** 3))
            if psingle_rayleigh == 1:
                p = ftau_cld * p + ftau_ray * (0.75 * (1 + ct ** 2.0))
        else:
            p = sum(wsg[l] * Pu0[l] * Pu1[l] for l in range(4))
        mus = (u1 + u0) / (u1 * u0)
        T = np.ones(nwno, dtype=dtau.dtype)
        kappa = np.zeros(nwno, dtype=dtau.dtype)
        zeta = R = delta = None
        for i in range(n):
            a = np.array([(2 * l + 1) - w0[i] * wmu[l, i] for l in range(4)])
            b = np.array([(F * (w0[i] * wsg[l, i])) * Pu0[l] / (4 * PI) for l in range(4)])
            lam1, lam2, Mn, Pl, E, Amat, beta, gama = sh4_layer(a, dtau[i])
            # particular solution (fluxes.py:3397-3416)
            x = 1 / u0
            Del = 9 * (x ** 4 - beta * x ** 2 + gama)
            a0, a1, a2, a3 = a
            b0, b1, b2, b3 = b
            D = [((a1 * b0 - b1 / u0) * (a2 * a3 - 9 / u0 ** 2) + 2 * (a3 * b2 - 2 * a3 * b0 - 3 * b3 / u0) / u0 ** 2),
                 ((a0 * b1 - b0 / u0) * (a2 * a3 - 9 / u0 ** 2) - 2 * a0 * (a3 * b2 - 3 * b3 / u0) / u0),
                 ((a3 * b2 - 3 * b3 / u0) * (a0 * a1 - 1 / u0 ** 2) - 2 * a3 * (a0 * b1 - b0 / u0) / u0),
                 ((a2 * b3 - 3 * b2 / u0) * (a0 * a1 - 1 / u0 ** 2) + 2 * (3 * a0 * b1 - 2 * a0 * b3 - 3 * b0 / u0) / u0 ** 2)]
            eta = [d / Del for d in D]
            zpl = np.stack([(eta[0] / 2 + eta[1] + 5 * eta[2] / 8) * 2 * PI,
                            (-eta[0] / 8 + 5 * eta[2] / 8 + eta[3]) * 2 * PI], axis=-1)
            zmn = np.stack([(eta[0] / 2 - eta[1] + 5 * eta[2] / 8) * 2 * PI,
                            (-eta[0] / 8 + 5 * eta[2] / 8 - eta[3]) * 2 * PI], axis=-1)
            ed = np.exp(-_clip(tau[i] / u0))[:, None]
            eu = np.exp(-_clip(tau[i + 1] / u0))[:, None]
            zmn_dn, zpl_dn, zmn_up, zpl_up = zmn * ed, zpl * ed, zmn * eu, zpl * eu
            ME = Mn * E[:, None, :]        # Mn @ diag(E)
            PE = Pl * E[:, None, :]
            # ---- functional weights of this layer ----
            cm = [sum(wmu[j, i] * Pu1[j] * Amat[j][m] for j in range(4)) for m in range(4)]
            al1, al2, be1, be2 = 1 / u1 + lam1, 1 / u1 + lam2, 1 / u1 - lam1, 1 / u1 - lam2
            h = lambda al: (1 - np.exp(-_clip(al * dtau[i]))) / al
            tw = T / u1 * w0[i]
            gd = np.stack([tw * cm[0] * h(al1), tw * cm[2] * h(al2)], axis=-1)
            gv = np.stack([tw * cm[1] * h(be1) * E[:, 0], tw * cm[3] * h(be2) * E[:, 1]], axis=-1)
            exptrm_mus = (1 - np.exp(-_clip(mus * dtau[i]))) / mus
            expon1 = exptrm_mus * np.exp(-_clip(tau[i] / u0))
            Nsum = sum(wmu[l, i] * Pu1[l] * eta[l] * expon1 for l in range(4))
            single = (w0_og[i] * F / (4 * PI) * p[i] * (1 - np.exp(-_clip(mus * dtau_og[i])))
                      * np.exp(-tau_og[i] / u0) / mus)
            c = T / u1 * (w0[i] * Nsum + single)
            Tn = T * np.exp(-dtau[i] / u1)
            if i == n - 1:      # xint[n] = flux_bot/pi = (Pl E d + Mn v + zpl_up)[0]/pi
                gd = gd + (Tn / PI)[:, None] * PE[:, 0, :]
                gv = gv + (Tn / PI)[:, None] * Mn[:, 0, :]
                c = c + Tn / PI * zpl_up[:, 0]
            if i == 0:
                Mni = _inv2(Mn)
                bt = np.stack([b_top - zmn_dn[:, 0], -b_top / 4 - zmn_dn[:, 1]], axis=-1)
                R = _mm(Mni, PE)
                delta = _mv(Mni, bt)
                kappa = c + np.einsum("...i,...i", gd, delta)
                zeta = gv - _mv(np.swapaxes(R, -1, -2), gd)
            else:
                A1 = pPl - _mm(pME, R)
                A2 = pMn - _mm(pPE, R)
                A2i = _inv2(A2)
                G = _mm(A1, A2i)
                cP = zpl_dn - p_zpl_up - _mv(pPE, delta)
                cM = _mv(pME, delta) + p_zmn_up - zmn_dn
                K = _inv2(Mn - _mm(G, Pl))
                Rn = -_mm(K, _mm(G, ME) - PE)
                deltan = _mv(K, _mv(G, cP) + cM)
                S = _mm(A2i, ME - _mm(Pl, Rn))
                t = _mv(A2i, _mv(Pl, deltan) + cP)
                kappa = kappa + np.einsum("...i,...i", zeta, t) + np.einsum("...i,...i", gd, deltan) + c
                zeta = _mv(np.swapaxes(S, -1, -2), zeta) + gv - _mv(np.swapaxes(Rn, -1, -2), gd)
                R, delta = Rn, deltan
            pMn, pPl, pME, pPE, p_zmn_up, p_zpl_up = Mn, Pl, ME, PE, zmn_up, zpl_up
            T = Tn
        bsf = rs * u0 * F * np.exp(-tau[n] / u0)
        bs = np.stack([bsf, -bsf / 4], axis=-1)
        rs2 = np.asarray(rs)[..., None, None] if np.ndim(rs) else rs
        lhs = (pMn - rs2 * pPl) - _mm(pPE - rs2 * pME, R)
        rhs = bs - p_zpl_up + (np.asarray(rs)[..., None] if np.ndim(rs) else rs) * p_zmn_up - _mv(pPE - rs2 * pME, delta)
        v = _mv(_inv2(lhs), rhs)
        out[k] = kappa + np.einsum("...i,...i", zeta, v)
    return out
