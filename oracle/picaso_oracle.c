/*
 * oracle/picaso_oracle.c -- TEST INFRASTRUCTURE, NOT THE PRODUCT.
 *
 * Plain-C, single-threaded, fp64 restatement of the reference PICASO per-wavelength
 * radiative-transfer hot path (natashabatalha/picaso v4.0.1, pure Python + numba).  It exists so
 * the HIP kernels can be checked on the GPU box, where the reference cannot travel.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Pinning: every function here is checked against golden vectors produced by importing the
 * reference's own source in the build container (tests/golden/make_golden.py -> the .npz fixtures under tests/golden;
 * tests/test_oracle_golden.py) and against the reference-owned Dlugach & Yanovitskij table
 * (reference/base_cases/testing/DLUGACH_TEST.csv via tests/test_dlugach.py).
 *
 * The operation order of each expression follows the reference so that agreement is ~1e-13
 * (the residual is libm-vs-numpy exp/sqrt last-ulp differences).  Layout everywhere is the
 * reference's: planes are C-order (nlayer|nlevel, nwno), wavelength contiguous.
 *
 * Each function cites the reference file:line it restates.
 */
#include <math.h>
#include <tgmath.h>
#include <stdlib.h>
#include <string.h>

/* The arithmetic type: double, or -- `make libpicaso_oracle_x80.so`, -DORC_REAL="long double" -- x87 extended
 * precision with the same expressions (<tgmath.h> picks expl / sqrtl / ...).  The extended build is how the tests
 * tell the reference's OWN fp64 rounding from an implementation error where its formulas are ill-conditioned
 * (thick layers: exp(+lambda dtau) terms up to the 35 clip): tests/helpers.py:lvl_excess, tests/test_fuzz_gpu.py.
 * Constants keep their fp64 values in both builds (the reference's `pi` is a Python float). */
#ifndef ORC_REAL
#define ORC_REAL double
#endif
typedef ORC_REAL real;

#define PI 3.141592653589793

/* ------------------------------------------------------------------------------------------
 * tri_diag_solve -- reference picaso/fluxes.py:288-323.
 * Thomas algorithm, eliminating from the LAST row upwards and substituting from the first row
 * downwards (this direction is what the reference does; kept for identical rounding).
 * as_, ds_ are caller-provided scratch of length l.
 * ------------------------------------------------------------------------------------------ */
static void tri_diag_solve(int l, const real *a, const real *b, const real *c,
                           const real *d, real *as_, real *ds_, real *xk)
{
    as_[l - 1] = a[l - 1] / b[l - 1];
    ds_[l - 1] = d[l - 1] / b[l - 1];
    for (int i = l - 2; i >= 0; --i) {
        real x = 1.0 / (b[i] - c[i] * as_[i + 1]);
        as_[i] = a[i] * x;
        ds_[i] = (d[i] - c[i] * ds_[i + 1]) * x;
    }
    xk[0] = ds_[0];
    for (int i = 1; i < l; ++i) xk[i] = ds_[i] - as_[i] * xk[i - 1];
}

/* ------------------------------------------------------------------------------------------
 * setup_tri_diag for ONE wavelength column -- reference picaso/fluxes.py:88-183.
 * Inputs are per-layer columns (length nlayer); outputs A,B,C,D of length 2*nlayer.
 * ------------------------------------------------------------------------------------------ */
static void setup_tri_diag_col(int nlayer, const real *c_plus_up, const real *c_minus_up,
                               const real *c_plus_down, const real *c_minus_down,
                               real b_top, real b_surface, real surf_reflect,
                               const real *gama, const real *ep, const real *em,
                               real *A, real *B, real *C, real *D)
{
    int L = 2 * nlayer;
    /* fluxes.py:143-146 */
#define E1(i) (ep[i] + gama[i] * em[i])
#define E2(i) (ep[i] - gama[i] * em[i])
#define E3(i) (gama[i] * ep[i] + em[i])
#define E4(i) (gama[i] * ep[i] - em[i])
    /* fluxes.py:155-158 */
    A[0] = 0.0;
    B[0] = gama[0] + 1.0;
    C[0] = gama[0] - 1.0;
    D[0] = b_top - c_minus_up[0];
    for (int i = 0; i < nlayer - 1; ++i) {
        /* rows 1,3,5,... fluxes.py:161-165 */
        int r = 2 * i + 1;
        A[r] = (E1(i) + E3(i)) * (gama[i + 1] - 1.0);
        B[r] = (E2(i) + E4(i)) * (gama[i + 1] - 1.0);
        C[r] = 2.0 * (1.0 - gama[i + 1] * gama[i + 1]);
        D[r] = ((gama[i + 1] - 1.0) * (c_plus_up[i + 1] - c_plus_down[i]) +
                (1.0 - gama[i + 1]) * (c_minus_down[i] - c_minus_up[i + 1]));
        /* rows 2,4,6,... fluxes.py:171-175 */
        r = 2 * i + 2;
        A[r] = 2.0 * (1.0 - gama[i] * gama[i]);
        B[r] = (E1(i) - E3(i)) * (gama[i + 1] + 1.0);
        C[r] = (E1(i) + E3(i)) * (gama[i + 1] - 1.0);
        D[r] = (E3(i) * (c_plus_up[i + 1] - c_plus_down[i]) +
                E1(i) * (c_minus_down[i] - c_minus_up[i + 1]));
    }
    /* fluxes.py:178-181 */
    int n = nlayer - 1;
    A[L - 1] = E1(n) - surf_reflect * E3(n);
    B[L - 1] = E2(n) - surf_reflect * E4(n);
    C[L - 1] = 0.0;
    D[L - 1] = b_surface - c_plus_down[n] + surf_reflect * c_minus_down[n];
#undef E1
#undef E2
#undef E3
#undef E4
}

static real hg_term(real g, real cos_theta)
{
    /* (1-g**2)/sqrt((1+g**2+2*g*cos_theta)**3)   fluxes.py:1316-1317 */
    real base = 1.0 + g * g + 2.0 * g * cos_theta;
    return (1.0 - g * g) / sqrt(base * base * base);
}

/* ------------------------------------------------------------------------------------------
 * Toon89 reflected light, 1-D or 3-D geometry.
 *   variant 0: get_reflected_1d  reference picaso/fluxes.py:1009-1413
 *   variant 1: get_reflected_3d  reference picaso/fluxes.py:354-660
 * Plane addressing: element (layer i, wave w) of facet (g,t) lives at
 *     p[(i*nwno + w)*fstride + foff],  fstride = 1 / foff = 0 for 1-D,
 *     fstride = numg*numt / foff = g*numt+t for 3-D  (reference layout (nlayer,nwno,ng,nt)).
 * surf_reflect and F0PI are (nwno) arrays (the scalar form is broadcast by the caller).
 * lvl (nullable): 4 arrays (numg,numt,nlevel,nwno): flux_minus, flux_plus, flux_minus_midpt,
 * flux_plus_midpt (only for variant 0 with get_lvl_flux).
 * ------------------------------------------------------------------------------------------ */
int orc_reflected(int variant, int nlevel, int nwno, int numg, int numt,
                  const real *dtau, const real *tau, const real *w0, const real *cosb,
                  const real *gcos2, const real *ftau_cld, const real *ftau_ray,
                  const real *dtau_og, const real *tau_og, const real *w0_og,
                  const real *cosb_og, const real *surf_reflect, const real *ubar0,
                  const real *ubar1, real cos_theta, const real *F0PI, int single_phase,
                  int multi_phase, real frac_a, real frac_b, real frac_c,
                  real constant_back, real constant_forward, int get_toa_intensity,
                  int get_lvl_flux, int toon_coefficients, real b_top_in, real *xint_at_top,
                  real *lvl_fm, real *lvl_fp, real *lvl_fmm, real *lvl_fpm)
{
    const int nlayer = nlevel - 1, L = 2 * nlayer;
    if (nlayer < 1 || nwno < 1) return 1;
    if (multi_phase != 0 && multi_phase != 1) return 2;      /* reference: UnboundLocalError */
    if (single_phase < 0 || single_phase > 3) return 2;
    if (toon_coefficients != 0 && toon_coefficients != 1) return 2;
    const real sq3 = sqrt(3.0);
    const int is3d = (variant == 1);
    const int nfac = numg * numt;
    const real clip = is3d ? 40.0 : 35.0;                   /* fluxes.py:1174 vs :516 */
    const real b_top = is3d ? 0.0 : b_top_in;               /* fluxes.py:522 */
    if (is3d) toon_coefficients = 0;                          /* fluxes.py:489-493 */

    size_t nl = (size_t)nlayer;
    real *buf = (real *)malloc(sizeof(real) * (nl * 22 + (size_t)L * 7 + 8));
    if (!buf) return 3;
    real *g1 = buf, *g2 = g1 + nl, *lam = g2 + nl, *gam = lam + nl, *cpu = gam + nl,
           *cmu = cpu + nl, *cpd = cmu + nl, *cmd = cpd + nl, *expt = cmd + nl, *ep = expt + nl,
           *em = ep + nl, *pos = em + nl, *neg = pos + nl, *apl = neg + nl, *ami = apl + nl,
           *Gq = ami + nl, *Hq = Gq + nl, *Aq = Hq + nl, *psing = Aq + nl;
    real *A = psing + nl * 4, *B = A + L, *C = B + L, *D = C + L, *AS = D + L, *DS = AS + L,
           *X = DS + L;

    for (int ig = 0; ig < numg; ++ig)
        for (int it = 0; it < numt; ++it) {
            const int fac = ig * numt + it;
            real u1 = ubar1[fac], u0 = ubar0[fac];
            if (is3d) { u1 = fabs(u1); u0 = fabs(u0); }        /* fluxes.py:467-468 */
            const size_t fs = is3d ? (size_t)nfac : 1, fo = is3d ? (size_t)fac : 0;
#define P(arr, i, w) arr[((size_t)(i) * nwno + (w)) * fs + fo]
            for (int w = 0; w < nwno; ++w) {
                const real F = F0PI[w], rs = surf_reflect[w];
                for (int i = 0; i < nlayer; ++i) {
                    const real w0_ = P(w0, i, w), cb = P(cosb, i, w), fc = P(ftau_cld, i, w);
                    if (toon_coefficients == 1) {              /* fluxes.py:1134-1135 */
                        g1[i] = (7.0 - w0_ * (4.0 + 3.0 * fc * cb)) / 4.0;
                        g2[i] = -(1.0 - w0_ * (4.0 - 3.0 * fc * cb)) / 4.0;
                    } else {                                   /* fluxes.py:1137-1138 */
                        g1[i] = (sq3 * 0.5) * (2.0 - w0_ * (1.0 + fc * cb));
                        g2[i] = (sq3 * w0_ * 0.5) * (1.0 - fc * cb);
                    }
                    lam[i] = sqrt(g1[i] * g1[i] - g2[i] * g2[i]);   /* :1140 */
                    gam[i] = (g1[i] - lam[i]) / g2[i];               /* :1141 */
                    real g3;
                    if (toon_coefficients == 1) g3 = (2.0 - 3.0 * fc * cb * u0) / 4.0;  /* :1149 */
                    else g3 = 0.5 * (1.0 - sq3 * fc * cb * u0);                         /* :1151 */
                    const real g4 = 1.0 - g3;
                    const real den = lam[i] * lam[i] - 1.0 / (u0 * u0);               /* :1155 */
                    ami[i] = F * w0_ * (g4 * (g1[i] + 1.0 / u0) + g2[i] * g3) / den;    /* :1158 */
                    apl[i] = F * w0_ * (g3 * (g1[i] - 1.0 / u0) + g2[i] * g4) / den;    /* :1159 */
                    real x = exp(-P(tau, i, w) / u0);                                  /* :1164 */
                    cmu[i] = ami[i] * x;
                    cpu[i] = apl[i] * x;
                    x = exp(-P(tau, i + 1, w) / u0);                                     /* :1167 */
                    cmd[i] = ami[i] * x;
                    cpd[i] = apl[i] * x;
                    real e = lam[i] * P(dtau, i, w);                                   /* :1172 */
                    if (e > clip) e = clip;                                              /* :1174 */
                    expt[i] = e;
                    ep[i] = exp(e);                                                      /* :1176 */
                    em[i] = 1.0 / ep[i];                                                 /* :1177 */
                }
                const real b_surface = 0.0 + rs * u0 * F * exp(-P(tau, nlayer, w) / u0); /* :1183 */
                setup_tri_diag_col(nlayer, cpu, cmu, cpd, cmd, b_top, b_surface, rs, gam, ep, em,
                                   A, B, C, D);
                tri_diag_solve(L, A, B, C, D, AS, DS, X);                                /* :1205 */
                for (int i = 0; i < nlayer; ++i) {                                       /* :1207-1208 */
                    pos[i] = X[2 * i] + X[2 * i + 1];
                    neg[i] = X[2 * i] - X[2 * i + 1];
                }
                const int n = nlayer - 1;
                if (get_lvl_flux && !is3d && lvl_fm) {                                   /* :1219-1257 */
                    size_t base = ((size_t)fac * nlevel) * nwno + w;
                    for (int i = 0; i < nlayer; ++i) {
                        real fm = pos[i] * gam[i] + neg[i] + cmu[i];
                        real fp = pos[i] + gam[i] * neg[i] + cpu[i];
                        fm = fm + u0 * F * exp(-P(tau, i, w) / u0);                      /* :1236 */
                        real epm = exp(0.5 * expt[i]);
                        real emm = 1.0 / epm;
                        real taumid = P(tau, i, w) + 0.5 * P(dtau, i, w);
                        real x = exp(-taumid / u0);
                        real cpm = apl[i] * x, cmm = ami[i] * x;
                        real fmm = gam[i] * pos[i] * epm + neg[i] * emm + cmm;
                        real fpm = pos[i] * epm + gam[i] * neg[i] * emm + cpm;
                        fmm = fmm + u0 * F * exp(-taumid / u0);                          /* :1251 */
                        lvl_fm[base + (size_t)i * nwno] = fm;
                        lvl_fp[base + (size_t)i * nwno] = fp;
                        lvl_fmm[base + (size_t)i * nwno] = fmm;
                        lvl_fpm[base + (size_t)i * nwno] = fpm;
                    }
                    real fzm = gam[n] * pos[n] * ep[n] + neg[n] * em[n] + cmd[n];      /* :1230 */
                    real fzp = pos[n] * ep[n] + gam[n] * neg[n] * em[n] + cpd[n];      /* :1231 */
                    fzm = fzm + u0 * F * exp(-P(tau, nlayer, w) / u0);
                    lvl_fm[base + (size_t)nlayer * nwno] = fzm;
                    lvl_fp[base + (size_t)nlayer * nwno] = fzp;
                    lvl_fmm[base + (size_t)nlayer * nwno] = 0.0;
                    lvl_fpm[base + (size_t)nlayer * nwno] = 0.0;
                }
                if (!get_toa_intensity && !is3d) continue;
                const real flux_zero = pos[n] * ep[n] + gam[n] * neg[n] * em[n] + cpd[n]; /* :1266 */
                real xint = flux_zero / PI;                                            /* :1270 */
                for (int i = 0; i < nlayer; ++i) {
                    const real w0_ = P(w0, i, w), cb = P(cosb, i, w), fc = P(ftau_cld, i, w);
                    real mp, mm;
                    if (multi_phase == 0) {                                              /* :1275-1284 */
                        const real ubar2 = 0.767;
                        const real q = P(gcos2, i, w) * (3.0 * ubar2 * ubar2 * u1 * u1 - 1.0) / 2.0;
                        mp = (1.0 + 1.5 * fc * cb * u1 + q);
                        mm = (1.0 - 1.5 * fc * cb * u1 + q);
                    } else {                                                             /* :1285-1287 */
                        mp = 1.0 + 1.5 * fc * cb * u1;
                        mm = 1.0 - 1.5 * fc * cb * u1;
                    }
                    /* :1290-1296 (3-D :581-587 multiplies w0 first; same value to 1 ulp) */
                    Gq[i] = pos[i] * (mp + gam[i] * mm) * w0_ * 0.5 / PI;
                    Hq[i] = neg[i] * (gam[i] * mp + mm) * w0_ * 0.5 / PI;
                    Aq[i] = (mp * cpu[i] + mm * cmu[i]) * w0_ * 0.5 / PI;
                    const real cbo = P(cosb_og, i, w);
                    real gf = 0, gb = 0, f = 0;
                    if (single_phase != 1) {                                             /* :1303-1306 */
                        gf = constant_forward * cbo;
                        gb = constant_back * cbo;
                        f = frac_a + frac_b * pow(gb, frac_c);
                    }
                    real p;
                    if (single_phase == 0) {
                        if (!is3d)                                                       /* :1315-1321 */
                            p = f * hg_term(gf, cos_theta) + (1.0 - f) * hg_term(gb, cos_theta) +
                                P(gcos2, i, w);
                        else {                                                           /* :605-615 */
                            real b1 = 1.0 + cbo * cbo + 2.0 * cbo * cos_theta;
                            real hb = -cbo / 2.0;
                            real b2 = 1.0 + hb * hb + 2.0 * hb * cos_theta;
                            p = f * (1.0 - gf * gf) / sqrt(b1 * b1 * b1) +
                                (1.0 - f) * (1.0 - gb * gb) / sqrt(b2 * b2 * b2) + P(gcos2, i, w);
                        }
                    } else if (single_phase == 1) {                                      /* :1324-1325 */
                        p = hg_term(cbo, cos_theta);
                    } else if (single_phase == 2) {                                      /* :1329-1333 */
                        p = f * hg_term(gf, cos_theta) + (1.0 - f) * hg_term(gb, cos_theta);
                    } else {                                                             /* :1338-1353 */
                        p = fc * (f * hg_term(gf, cos_theta) + (1.0 - f) * hg_term(gb, cos_theta)) +
                            P(ftau_ray, i, w) * (0.75 * (1.0 + cos_theta * cos_theta));
                    }
                    psing[i] = p;
                }
                for (int i = nlayer - 1; i >= 0; --i) {                                  /* :1381-1407 */
                    const real dt = P(dtau, i, w);
                    xint = (xint * exp(-dt / u1) +
                            (P(w0_og, i, w) * F / (4.0 * PI)) * psing[i] * exp(-P(tau_og, i, w) / u0) *
                                (1.0 - exp(-P(dtau_og, i, w) * (u0 + u1) / (u0 * u1))) *
                                (u0 / (u0 + u1)) +
                            Aq[i] * (1.0 - exp(-dt * (u0 + 1 * u1) / (u0 * u1))) * (u0 / (u0 + 1 * u1)) +
                            Gq[i] * (exp(expt[i] * 1 - dt / u1) - 1.0) / (lam[i] * 1 * u1 - 1.0) +
                            Hq[i] * (1.0 - exp(-expt[i] * 1 - dt / u1)) / (lam[i] * 1 * u1 + 1.0));
                }
                xint_at_top[(size_t)fac * nwno + w] = xint;                              /* :1410 */
            }
#undef P
        }
    free(buf);
    return 0;
}

/* blackbody -- reference picaso/fluxes.py:1660-1680 (cgs, per unit wavelength, w in cm) */
static real planck_lambda(real t, real wcm)
{
    const real h = 6.62607004e-27, c = 2.99792458e+10, k = 1.38064852e-16;
    return ((2.0 * h * (c * c)) / pow(wcm, 5.0)) * (1.0 / (exp((h * c) / (t * (wcm * k))) - 1.0));
}

/* blackbody_integrated -- reference picaso/fluxes.py:1608-1658 (3-point bin mean in wavenumber) */
static real planck_integrated(real t, real wave, real dwave)
{
    const real h = 6.62607004e-27, c = 2.99792458e+10, k = 1.38064852e-16;
    const real c1 = 2 * h * (c * c), c2 = h * c / k;
    const int nbb = 1;
    real s = 0.0;
    for (int kk = -nbb; kk <= nbb; ++kk) {
        real wn = wave + kk * dwave / (2.0 * nbb);
        s += c1 * (wn * wn * wn) / (exp(c2 * wn / t) - 1.0);
    }
    return s / (2 * nbb + 1.0);
}

/* ------------------------------------------------------------------------------------------
 * Toon89 thermal emission, 1-D or 3-D.
 *   variant 0: get_thermal_1d  reference picaso/fluxes.py:1682-1912
 *   variant 1: get_thermal_3d  reference picaso/fluxes.py:2147-2352
 * 1-D: tlevel, plevel are (nlevel); planes (nlayer,nwno).  3-D: tlevel/plevel are
 * (nlevel,ng,nt), planes (nlayer,nwno,ng,nt).  lvl_* (nullable, 1-D only): 4 arrays
 * (numg,numt,nlevel,nwno) always filled by the reference.
 * ------------------------------------------------------------------------------------------ */
int orc_thermal(int variant, int nlevel, const real *wno, int nwno, int numg, int numt,
                const real *tlevel, const real *dtau, const real *w0, const real *cosb,
                const real *plevel, const real *ubar1, const real *surf_reflect,
                int hard_surface, const real *dwno, int calc_type, real *flux_at_top,
                real *lvl_fm, real *lvl_fp, real *lvl_fmm, real *lvl_fpm)
{
    const int nlayer = nlevel - 1, L = 2 * nlayer;
    if (nlayer < 1 || nwno < 1) return 1;
    const int is3d = (variant == 1);
    const int nfac = numg * numt;
    const real mu1 = 0.5;                                                      /* :1748 */
    size_t nl = (size_t)nlayer;
    real *buf = (real *)malloc(sizeof(real) * (nl * 30 + (size_t)nlevel * 6 + (size_t)L * 7 + 8));
    if (!buf) return 3;
    real *allb = buf, *b0 = allb + nlevel, *b1 = b0 + nl, *g1 = b1 + nl, *g2 = g1 + nl,
           *lam = g2 + nl, *gam = lam + nl, *gpg = gam + nl, *cpu = gpg + nl, *cmu = cpu + nl,
           *cpd = cmu + nl, *cmd = cpd + nl, *expt = cmd + nl, *ep = expt + nl, *em = ep + nl,
           *pos = em + nl, *neg = pos + nl, *Gq = neg + nl, *Hq = Gq + nl, *Jq = Hq + nl,
           *Kq = Jq + nl, *al1 = Kq + nl, *al2 = al1 + nl, *si1 = al2 + nl, *si2 = si1 + nl,
           *epm = si2 + nl, *emm = epm + nl;
    real *fmn = emm + nl, *fpl = fmn + nlevel, *fmm = fpl + nlevel, *fpm = fmm + nlevel;
    real *A = fpm + nlevel + nl, *B = A + L, *C = B + L, *D = C + L, *AS = D + L, *DS = AS + L,
           *X = DS + L;

    const int nouter = is3d ? nfac : 1;     /* 3-D redoes the solve per facet (:2213-2214) */
    for (int fo_ = 0; fo_ < nouter; ++fo_) {
        const size_t fs = is3d ? (size_t)nfac : 1, fo = is3d ? (size_t)fo_ : 0;
#define P(arr, i, w) arr[((size_t)(i) * nwno + (w)) * fs + fo]
#define LV(arr, i) arr[(size_t)(i) * fs + fo]
        for (int w = 0; w < nwno; ++w) {
            const real rs = surf_reflect[w];
            for (int l = 0; l < nlevel; ++l) {
                if (calc_type == 0 || is3d) allb[l] = planck_lambda(LV(tlevel, l), 1.0 / wno[w]); /* :1752 */
                else allb[l] = planck_integrated(LV(tlevel, l), wno[w], dwno[w]);                /* :1754 */
            }
            for (int i = 0; i < nlayer; ++i) {
                const real dt = P(dtau, i, w), w0_ = P(w0, i, w), cb = P(cosb, i, w);
                b0[i] = allb[i];                                                  /* :1756 */
                b1[i] = (allb[i + 1] - b0[i]) / dt;                               /* :1757 */
                g1[i] = 2.0 - w0_ * (1 + cb);                                     /* :1760 */
                g2[i] = w0_ * (1 - cb);
                lam[i] = sqrt(g1[i] * g1[i] - g2[i] * g2[i]);                     /* :1763 */
                gam[i] = (g1[i] - lam[i]) / g2[i];                                /* :1764 */
                gpg[i] = 1.0 / (g1[i] + g2[i]);                                   /* :1766 */
                cpu[i] = 2 * PI * mu1 * (b0[i] + b1[i] * gpg[i]);                 /* :1772 */
                cmu[i] = 2 * PI * mu1 * (b0[i] - b1[i] * gpg[i]);                 /* :1773 */
                cpd[i] = 2 * PI * mu1 * (b0[i] + b1[i] * dt + b1[i] * gpg[i]);    /* :1778 */
                cmd[i] = 2 * PI * mu1 * (b0[i] + b1[i] * dt - b1[i] * gpg[i]);    /* :1779 */
                real e = lam[i] * dt;                                           /* :1784 */
                if (e > 35.0) e = 35.0;                                           /* :1786 */
                expt[i] = e;
                ep[i] = exp(e);
                em[i] = 1.0 / ep[i];
            }
            const real tau_top = P(dtau, 0, w) * LV(plevel, 0) / (LV(plevel, 1) - LV(plevel, 0)); /* :1797 */
            real b_top, b_surface;
            if (!is3d) {
                b_top = (1.0 - exp(-tau_top / mu1)) * allb[0] * PI;                /* :1800 */
                if (hard_surface) b_surface = (1.0 - rs) * allb[nlevel - 1] * PI;  /* :1803-1804 */
                else b_surface = (allb[nlevel - 1] + b1[nlayer - 1] * mu1) * PI;   /* :1806 */
            } else {
                b_top = PI * (1.0 - exp(-tau_top / mu1)) * allb[0];                /* :2253 */
                if (hard_surface) b_surface = PI * allb[nlevel - 1];               /* :2256 */
                else b_surface = PI * (allb[nlevel - 1] + b1[nlayer - 1] * mu1);   /* :2258 */
            }
            setup_tri_diag_col(nlayer, cpu, cmu, cpd, cmd, b_top, b_surface, rs, gam, ep, em, A, B,
                               C, D);
            tri_diag_solve(L, A, B, C, D, AS, DS, X);
            for (int i = 0; i < nlayer; ++i) {
                pos[i] = X[2 * i] + X[2 * i + 1];
                neg[i] = X[2 * i] - X[2 * i + 1];
                Gq[i] = (1 / mu1 - lam[i]) * pos[i];                               /* :1842-1849 */
                Hq[i] = gam[i] * (lam[i] + 1 / mu1) * neg[i];
                Jq[i] = gam[i] * (lam[i] + 1 / mu1) * pos[i];
                Kq[i] = (1 / mu1 - lam[i]) * neg[i];
                al1[i] = 2 * PI * (b0[i] + b1[i] * (gpg[i] - mu1));
                al2[i] = 2 * PI * b1[i];
                si1[i] = 2 * PI * (b0[i] - b1[i] * (gpg[i] - mu1));
                si2[i] = 2 * PI * b1[i];
                epm[i] = exp(0.5 * expt[i]);                                       /* :1856 */
                emm[i] = 1 / epm[i];
            }
            const int a_lo = is3d ? fo_ : 0, a_hi = is3d ? fo_ + 1 : nfac;
            for (int fac = a_lo; fac < a_hi; ++fac) {
                const real iu = ubar1[fac];
                memset(fmn, 0, sizeof(real) * 4 * (size_t)nlevel);
                if (!is3d) {
                    if (hard_surface) fpl[nlevel - 1] = (1.0 - rs) * allb[nlevel - 1] * 2 * PI;   /* :1871 */
                    else fpl[nlevel - 1] = (allb[nlevel - 1] + b1[nlayer - 1] * iu) * 2 * PI;      /* :1873 */
                    fmn[0] = (1 - exp(-tau_top / iu)) * allb[0] * 2 * PI;                          /* :1875 */
                } else {
                    if (hard_surface) fpl[nlevel - 1] = PI * (b_surface);                          /* :2310 */
                    else fpl[nlevel - 1] = PI * (allb[nlevel - 1] + b1[nlayer - 1] * iu);          /* :2312 */
                    fmn[0] = PI * (1 - exp(-tau_top / iu)) * allb[0];                              /* :2315 */
                }
                for (int itop = 0; itop < nlayer; ++itop) {                                        /* :1880-1907 */
                    {
                        const int i = itop;
                        const real dt = P(dtau, i, w);
                        const real ea = exp(-dt / iu), eam = exp(-0.5 * dt / iu);
                        fmn[i + 1] = (fmn[i] * ea + (Jq[i] / (lam[i] * iu + 1.0)) * (ep[i] - ea) +
                                      (Kq[i] / (lam[i] * iu - 1.0)) * (ea - em[i]) + si1[i] * (1. - ea) +
                                      si2[i] * (iu * ea + dt - iu));
                        fmm[i] = (fmn[i] * eam + (Jq[i] / (lam[i] * iu + 1.0)) * (epm[i] - eam) +
                                  (Kq[i] / (-lam[i] * iu + 1.0)) * (emm[i] - eam) + si1[i] * (1. - eam) +
                                  si2[i] * (iu * eam + 0.5 * dt - iu));
                    }
                    {
                        const int i = nlayer - 1 - itop;
                        const real dt = P(dtau, i, w);
                        const real ea = exp(-dt / iu), eam = exp(-0.5 * dt / iu);
                        fpl[i] = (fpl[i + 1] * ea + (Gq[i] / (lam[i] * iu - 1.0)) * (ep[i] * ea - 1.0) +
                                  (Hq[i] / (lam[i] * iu + 1.0)) * (1.0 - em[i] * ea) + al1[i] * (1. - ea) +
                                  al2[i] * (iu - (dt + iu) * ea));
                        fpm[i] = (fpl[i + 1] * eam + (Gq[i] / (lam[i] * iu - 1.0)) * (ep[i] * eam - epm[i]) -
                                  (Hq[i] / (lam[i] * iu + 1.0)) * (em[i] * eam - emm[i]) + al1[i] * (1. - eam) +
                                  al2[i] * (iu + 0.5 * dt - (dt + iu) * eam));
                    }
                }
                flux_at_top[(size_t)fac * nwno + w] = fpm[0];                                      /* :1910 */
                if (!is3d && lvl_fm) {
                    size_t base = ((size_t)fac * nlevel) * nwno + w;
                    for (int l = 0; l < nlevel; ++l) {
                        lvl_fm[base + (size_t)l * nwno] = fmn[l];
                        lvl_fp[base + (size_t)l * nwno] = fpl[l];
                        lvl_fmm[base + (size_t)l * nwno] = fmm[l];
                        lvl_fpm[base + (size_t)l * nwno] = fpm[l];
                    }
                }
            }
        }
#undef P
#undef LV
    }
    free(buf);
    return 0;
}

/* compress_disco -- reference picaso/disco.py:117-149 */
int orc_compress_disco(int nwno, real cos_theta, const real *xint_at_top, const real *gweight,
                       int ng, const real *tweight, int nt, const real *F0PI, real *albedo)
{
    const real sym_fac = (nt == 1) ? 2 * PI : 1.0;
    for (int w = 0; w < nwno; ++w) {
        real a = 0.0;
        for (int ig = 0; ig < ng; ++ig)
            for (int it = 0; it < nt; ++it)
                a = a + xint_at_top[((size_t)ig * nt + it) * nwno + w] * gweight[ig] * tweight[it];
        albedo[w] = sym_fac * 0.5 * a / F0PI[w] * (cos_theta + 1.0);
    }
    return 0;
}

/* compress_thermal -- reference picaso/disco.py:151-181; ninner = nwno (3-D input) or
 * nlevel*nwno (4-D input) */
int orc_compress_thermal(size_t ninner, const real *flux_at_top, const real *gweight, int ng,
                         const real *tweight, int nt, real *flux)
{
    const real sym_fac = (nt == 1) ? 1.0 : 1.0 / (2 * PI);
    for (size_t w = 0; w < ninner; ++w) {
        real a = 0.0;
        for (int ig = 0; ig < ng; ++ig)
            for (int it = 0; it < nt; ++it)
                a = a + flux_at_top[((size_t)ig * nt + it) * ninner + w] * gweight[ig] * tweight[it];
        flux[w] = a * sym_fac;
    }
    return 0;
}

/* get_transit_1d -- reference picaso/fluxes.py:2581-2663 (Brown 2001 eq. 11).  Loop structure and
 * operation order of the reference: delta_length[i][j] (:2625-2644), TAU = DTAU/colden*mmw
 * (:2648-2650), TAUALL accumulated in j order (:2653-2656), F (:2660-2661).  player/tlayer are
 * indexed exactly as the reference indexes them (arrays of length >= nlevel-1). */
int orc_get_transit_1d(const real *z, const real *dz, int nlevel, int nwno, real rstar,
                       const real *mmw, real k_b, real amu, const real *player,
                       const real *tlayer, const real *colden, const real *DTAU, real *F)
{
    const int n = nlevel, nl = nlevel - 1;
    real *dlen = (real *)calloc((size_t)n * n, sizeof(real));
    real *tau = (real *)malloc(sizeof(real) * nl);
    if (!dlen || !tau) { free(dlen); free(tau); return 1; }
    real seg = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) {
            const real ref = z[i], inner = z[i - j], outer = z[i - j - 1];
            if (inner != ref && outer != ref)
                seg = sqrt(outer * outer - ref * ref) - sqrt(inner * inner - ref * ref);
            else if (inner == ref)
                seg = sqrt(outer * outer - ref * ref);
            dlen[(size_t)i * n + j] = seg * player[i - j - 1] / tlayer[i - j - 1] / k_b;
        }
    real zmin = z[0];
    for (int i = 1; i < n; ++i) zmin = z[i] < zmin ? z[i] : zmin;
    for (int w = 0; w < nwno; ++w) {
        for (int l = 0; l < nl; ++l) tau[l] = DTAU[(size_t)l * nwno + w] / colden[l] * (mmw[l] * amu);
        real acc = 0.0;
        for (int i = 0; i < n; ++i) {
            real t = 0.0;
            for (int j = 0; j < i; ++j) t = t + 2 * tau[i - j - 1] * dlen[(size_t)i * n + j];
            acc = acc + (1.0 - exp(-t)) * (z[i] * dz[i]);
        }
        F[w] = (zmin / rstar) * (zmin / rstar) + 2. / (rstar * rstar) * acc;
    }
    free(dlen);
    free(tau);
    return 0;
}
