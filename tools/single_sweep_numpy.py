"""numpy statement of the single top-down sweep the HIP kernels implement (any float dtype).

Development / test aid: lets the algorithm in picaso_amd/csrc/toon_reflected.hip be checked on a
CPU (tests/test_single_sweep_numpy.py) and evaluated in extended precision (np.longdouble) to
judge which of two fp64 results is closer to the exact answer for given inputs.  Not used by the
product path.
"""
import numpy as np

PI = np.pi


def _hg(g, ct):
    b = 1 + g * g + 2 * g * ct
    return (1 - g * g) / np.sqrt(b * b * b)


def reflected_toa(nlevel, nwno, planes, rs, u0a, u1a, ct, F, sp, mp, fa, fb, fc_, cb_, cf_, tc,
                  b_top, clip=35.0):
    """Toon reflected TOA intensity, unknowns (pos,neg); relation neg_i = delta_i - rho_i pos_i and
    functional J = kappa + zeta pos_i carried downwards (see toon_reflected.hip header)."""
    dtau, tau, w0, cosb, gcos2, ftc, ftr, dtau_og, tau_og, w0_og, cosb_og = planes
    n = nlevel - 1
    dt_ = dtau.dtype
    sq3 = np.sqrt(dt_.type(3.0))
    out = np.zeros((len(u0a), nwno), dtype=dt_)
    for k, (u0, u1) in enumerate(zip(u0a, u1a)):
        T = np.ones(nwno, dtype=dt_)
        const = 0
        for i in range(n):
            w, g, fc = w0[i], cosb[i], ftc[i]
            if tc == 1:
                g1 = (7 - w * (4 + 3 * fc * g)) / 4
                g2 = -(1 - w * (4 - 3 * fc * g)) / 4
                g3 = (2 - 3 * fc * g * u0) / 4
            else:
                g1 = (sq3 * .5) * (2 - w * (1 + fc * g))
                g2 = (sq3 * w * .5) * (1 - fc * g)
                g3 = .5 * (1 - sq3 * fc * g * u0)
            lam = np.sqrt(g1 * g1 - g2 * g2)
            gam = (g1 - lam) / g2
            g4 = 1 - g3
            den = lam * lam - 1 / (u0 * u0)
            am = F * w * (g4 * (g1 + 1 / u0) + g2 * g3) / den
            ap = F * w * (g3 * (g1 - 1 / u0) + g2 * g4) / den
            xu = np.exp(-tau[i] / u0)
            xd = np.exp(-tau[i + 1] / u0)
            cmu, cpu, cmd, cpd = am * xu, ap * xu, am * xd, ap * xd
            E = np.minimum(lam * dtau[i], clip)
            EP = np.exp(E)
            EM = 1 / EP
            q = gcos2[i] * (3 * 0.767 * 0.767 * u1 * u1 - 1) / 2 if mp == 0 else 0
            mpl = 1 + 1.5 * fc * g * u1 + q
            mmi = 1 - 1.5 * fc * g * u1 + q
            et = np.exp(-dtau[i] / u1)
            gc = w * (mpl + gam * mmi) * 0.5 / PI * (np.exp(E - dtau[i] / u1) - 1) / (lam * u1 - 1)
            hc = w * (gam * mpl + mmi) * 0.5 / PI * (1 - np.exp(-E - dtau[i] / u1)) / (lam * u1 + 1)
            Aq = (mpl * cpu + mmi * cmu) * w * 0.5 / PI
            cbo = cosb_og[i]
            if sp != 1:
                gf, gb = cf_ * cbo, cb_ * cbo
                f = fa + fb * gb ** fc_
            if sp == 0:
                p = f * _hg(gf, ct) + (1 - f) * _hg(gb, ct) + gcos2[i]
            elif sp == 1:
                p = _hg(cbo, ct)
            elif sp == 2:
                p = f * _hg(gf, ct) + (1 - f) * _hg(gb, ct)
            else:
                p = fc * (f * _hg(gf, ct) + (1 - f) * _hg(gb, ct)) + ftr[i] * (0.75 * (1 + ct * ct))
            mus = (u0 + u1) / (u0 * u1)
            S0 = ((w0_og[i] * F / (4 * PI)) * p * np.exp(-tau_og[i] / u0)
                  * (1 - np.exp(-dtau_og[i] * mus)) * (u0 / (u0 + u1))
                  + Aq * (1 - np.exp(-dtau[i] * mus)) * (u0 / (u0 + u1)))
            const = const + T * S0
            Tn = T * et
            vp, vn = T * gc, T * hc
            if i == n - 1:
                vp = vp + Tn * EP / PI
                vn = vn + Tn * gam * EM / PI
                const = const + Tn * cpd / PI
            if i == 0:
                rho, delta = gam, b_top - cmu
                zeta, kappa = vp - vn * rho, vn * delta
            else:
                em2 = p_EM * p_EM
                a1 = 1 - p_gam * em2 * rho
                a2 = p_gam - em2 * rho
                inv = 1 / (a1 - gam * a2)
                rP = (cpu - p_cpd) - p_gam * p_EM * delta
                rM = (cmu - p_cmd) - p_EM * delta
                rho_n = (gam * a1 - a2) * inv
                delta_n = (a2 * rP - a1 * rM) * inv
                ia = p_EM / a1
                s = (1 - gam * rho_n) * ia
                t = (gam * delta_n + rP) * ia
                kappa = kappa + zeta * t + vn * delta_n
                zeta = zeta * s + vp - vn * rho_n
                rho, delta = rho_n, delta_n
            p_gam, p_EM, p_cpd, p_cmd = gam, EM, cpd, cmd
            T = Tn
        bs = rs * u0 * F * np.exp(-tau[n] / u0)
        em2 = p_EM * p_EM
        pos = ((p_EM * (bs - p_cpd + rs * p_cmd) - em2 * (p_gam - rs) * delta)
               / ((1 - rs * p_gam) - em2 * (p_gam - rs) * rho))
        out[k] = const + kappa + zeta * pos
    return out


def planck(t, wno):
    h, c, k = 6.62607004e-27, 2.99792458e+10, 1.38064852e-16
    w = 1 / wno
    return ((2.0 * h * c ** 2.0) / (w ** 5.0)) * (1.0 / (np.exp((h * c) / (t * (w * k))) - 1.0))


def thermal_toa(nlevel, wno, nwno, tlevel, dtau, w0, cosb, plevel, u1a, rs, hard):
    """Toon thermal flux_at_top = F+m[0] by the same sweep (see toon_thermal.hip header)."""
    n = nlevel - 1
    mu1 = 0.5
    na = len(u1a)
    out = np.zeros((na, nwno), dtype=dtau.dtype)
    Bn = planck(tlevel[0], wno)
    tau_top = dtau[0] * plevel[0] / (plevel[1] - plevel[0])
    b_top = (1 - np.exp(-tau_top / mu1)) * Bn * PI
    W, kappa, zeta = [None] * na, [0] * na, [None] * na
    for i in range(n):
        B0 = Bn
        Bn = planck(tlevel[i + 1], wno)
        dt, w, g = dtau[i], w0[i], cosb[i]
        b1 = (Bn - B0) / dt
        g1 = 2 - w * (1 + g)
        g2 = w * (1 - g)
        lam = np.sqrt(g1 * g1 - g2 * g2)
        gam = (g1 - lam) / g2
        s = 1 / (g1 + g2)
        cpu = 2 * PI * mu1 * (B0 + b1 * s)
        cmu = 2 * PI * mu1 * (B0 - b1 * s)
        cpd = 2 * PI * mu1 * (B0 + b1 * dt + b1 * s)
        cmd = 2 * PI * mu1 * (B0 + b1 * dt - b1 * s)
        E = np.minimum(lam * dt, 35.)
        EP = np.exp(E)
        EM = 1 / EP
        al1 = 2 * PI * (B0 + b1 * (s - mu1))
        al2 = 2 * PI * b1
        if i == 0:
            rho, delta = gam, b_top - cmu
        else:
            em2 = p_EM * p_EM
            a1 = 1 - p_gam * em2 * rho
            a2 = p_gam - em2 * rho
            inv = 1 / (a1 - gam * a2)
            rP = (cpu - p_cpd) - p_gam * p_EM * delta
            rM = (cmu - p_cmd) - p_EM * delta
            rho_n = (gam * a1 - a2) * inv
            delta_n = (a2 * rP - a1 * rM) * inv
            ia = p_EM / a1
            sfac = (1 - gam * rho_n) * ia
            t = (gam * delta_n + rP) * ia
        for k, mu in enumerate(u1a):
            e = np.exp(-dt / mu)
            if i == 0:
                EPm = np.exp(0.5 * E)
                EMm = 1 / EPm
                em = np.exp(-0.5 * dt / mu)
                vp = (2 - lam) / (lam * mu - 1) * (EP * em - EPm)
                vn = -gam * (lam + 2) / (lam * mu + 1) * (EM * em - EMm)
                c0 = al1 * (1 - em) + al2 * (mu + 0.5 * dt - (dt + mu) * em)
                kappa[k] = c0 + vn * delta
                zeta[k] = vp - vn * rho
                W[k] = em
            else:
                vp = W[k] * (2 - lam) / (lam * mu - 1) * (EP * e - 1)
                vn = W[k] * gam * (lam + 2) / (lam * mu + 1) * (1 - EM * e)
                c0 = W[k] * (al1 * (1 - e) + al2 * (mu - (dt + mu) * e))
                kappa[k] = kappa[k] + c0 + zeta[k] * t + vn * delta_n
                zeta[k] = zeta[k] * sfac + vp - vn * rho_n
                W[k] = W[k] * e
            if i == n - 1:
                fb = (1 - rs) * Bn * 2 * PI if hard else (Bn + b1 * mu) * 2 * PI
                kappa[k] = kappa[k] + W[k] * fb
        if i > 0:
            rho, delta = rho_n, delta_n
        p_gam, p_EM, p_cpd, p_cmd = gam, EM, cpd, cmd
        last_b1 = b1
    bs = (1 - rs) * Bn * PI if hard else (Bn + last_b1 * mu1) * PI
    em2 = p_EM * p_EM
    pos = ((p_EM * (bs - p_cpd + rs * p_cmd) - em2 * (p_gam - rs) * delta)
           / ((1 - rs * p_gam) - em2 * (p_gam - rs) * rho))
    for k in range(na):
        out[k] = kappa[k] + zeta[k] * pos
    return out
