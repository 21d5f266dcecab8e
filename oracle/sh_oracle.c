/*
 * oracle/sh_oracle.c -- TEST INFRASTRUCTURE, NOT THE PRODUCT.
 *
 * Plain-C restatement of the reference's spherical-harmonics (SH2 / SH4) solvers:
 *   get_reflected_SH        reference picaso/fluxes.py:2675-2976
 *   get_thermal_SH          reference picaso/fluxes.py:2979-3186   (flx = 0 only: flx = 1 is broken
 *                                                                   in the reference, :3102)
 *   setup_2_stream_fluxes   reference picaso/fluxes.py:3189-3333
 *   setup_4_stream_fluxes   reference picaso/fluxes.py:3336-3607
 *   solve_4_stream_banded   reference picaso/fluxes.py:3610-3628 = scipy.linalg.solve_banded =
 *                           LAPACK dgbsv: unblocked banded LU with partial pivoting (dgbtf2, first
 *                           maximal |pivot| like idamax) + dgbtrs.  scipy is a third-party dependency
 *                           of the reference (pyproject.toml pins no version; 1.15.3 in the build
 *                           container); its algorithm is restated in gb_solve() below.
 * Pinned against golden vectors produced by the reference's own source
 * (tests/golden/make_golden.py sh -> tests/golden/scene_sh*.npz; tests/test_oracle_golden.py).
 *
 * Layout as everywhere: planes (nlayer|nlevel, nwno) C-order.  The dense band for one wavelength is
 * built column-major like LAPACK band storage AB(kl+ku+1+kl, n).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PI 3.141592653589793

static double clip35(double x) { return x > 35.0 ? 35.0 : (x < -35.0 ? -35.0 : x); }   /* slice_rav :68-76 */

static void legP(double mu, double *P)   /* fluxes.py:3639-3646 */
{
    P[0] = 1;
    P[1] = mu;
    P[2] = (3 * mu * mu - 1) / 2;
    P[3] = (5 * mu * mu * mu - 3 * mu) / 2;
    P[4] = (35 * pow(mu, 4) - 30 * mu * mu + 3) / 8;
    P[5] = (63 * pow(mu, 5) - 70 * mu * mu * mu + 15 * mu) / 8;
    P[6] = (231 * pow(mu, 6) - 315 * pow(mu, 4) + 105 * mu * mu - 5) / 16;
}

/* LAPACK dgbsv restatement.  ab: (ldab = 2*kl+ku+1) x n column-major band storage with the matrix in
 * rows kl..2kl+ku (0-based), i.e. ab[kl+ku+i-j + j*ldab] = M[i][j].  b overwritten with the solution. */
static int gb_solve(int n, int kl, int ku, double *ab, int ldab, double *b, int *ipiv)
{
    const int kv = ku + kl;
    /* dgbtf2 */
    for (int j = ku + 1; j < (kv < n ? kv : n); ++j)
        for (int i = kv - j + 1; i < kl; ++i) ab[i + j * ldab] = 0.0;
    int ju = 0;
    for (int j = 0; j < n; ++j) {
        if (j + kv < n)
            for (int i = 0; i < kl; ++i) ab[i + (j + kv) * ldab] = 0.0;
        const int km = (kl < n - 1 - j) ? kl : n - 1 - j;
        int jp = 0;
        double amax = fabs(ab[kv + j * ldab]);
        for (int i = 1; i <= km; ++i) {
            const double v = fabs(ab[kv + i + j * ldab]);
            if (v > amax) { amax = v; jp = i; }
        }
        ipiv[j] = jp + j;
        if (ab[kv + jp + j * ldab] != 0.0) {
            int t = j + ku + jp;
            if (t > n - 1) t = n - 1;
            if (t > ju) ju = t;
            if (jp != 0)
                for (int c = j; c <= ju; ++c) {   /* dswap of the rows within the band */
                    const double tmp = ab[kv + jp - (c - j) + c * ldab];
                    ab[kv + jp - (c - j) + c * ldab] = ab[kv - (c - j) + c * ldab];
                    ab[kv - (c - j) + c * ldab] = tmp;
                }
            if (km > 0) {
                const double r = 1.0 / ab[kv + j * ldab];
                for (int i = 1; i <= km; ++i) ab[kv + i + j * ldab] *= r;
                if (ju > j)
                    for (int c = j + 1; c <= ju; ++c) {   /* dger rank-1 update */
                        const double y = ab[kv - (c - j) + c * ldab];
                        if (y != 0.0)
                            for (int i = 1; i <= km; ++i)
                                ab[kv + i - (c - j) + c * ldab] -= ab[kv + i + j * ldab] * y;
                    }
            }
        } else
            return j + 1;
    }
    /* dgbtrs, no transpose: L then U */
    for (int j = 0; j < n - 1; ++j) {
        const int lm = (kl < n - 1 - j) ? kl : n - 1 - j;
        const int l = ipiv[j];
        if (l != j) { const double t = b[l]; b[l] = b[j]; b[j] = t; }
        for (int i = 1; i <= lm; ++i) b[j + i] -= ab[kv + i + j * ldab] * b[j];
    }
    for (int j = n - 1; j >= 0; --j) {   /* dtbsv upper, non-unit */
        if (b[j] != 0.0) {
            b[j] /= ab[kv + j * ldab];
            const int lo = (j - kv > 0) ? j - kv : 0;
            for (int i = j - 1; i >= lo; --i) b[i] -= b[j] * ab[kv - (j - i) + j * ldab];
        }
    }
    return 0;
}

#define MB(d, col) ab[(kl + (d)) + (size_t)(col) * ldab]   /* Mb[d, col] of the reference */

/* One wavelength of setup_4_stream_fluxes + solve.  Per-layer inputs are columns of length n. */
typedef struct {
    int n;
    double *lam1, *lam2, *eta /*4n*/, *A /*16n: A[j][m][i]*/, *X /*4n*/;
    double flux_bot;
    double *flux;       /* nullable: 4(n+1) moment fluxes F.X + G (fluxes=1, :3552-3599) */
} sh4_out;

static int sh4_column(int n, const double *w0, const double *dtau, const double *tau /*n+1*/, const double *a /*4n*/,
                      const double *bb /*4n*/, double b_top, double b_surface, double b_surface_SH4,
                      double rs, double ubar0, const double *B0, const double *B1, int calculation,
                      double *ab, int *ipiv, double *rhs, sh4_out *o)
{
    const int kl = 5, ku = 5, ldab = 2 * kl + ku + 1, N = 4 * n;
    memset(ab, 0, sizeof(double) * (size_t)ldab * N);
    memset(rhs, 0, sizeof(double) * N);
    double *p1pl = (double *)malloc(sizeof(double) * n * 40);
    double *p2pl = p1pl + n, *q1pl = p2pl + n, *q2pl = q1pl + n, *p1mn = q2pl + n, *p2mn = p1mn + n,
           *q1mn = p2mn + n, *q2mn = q1mn + n, *f = q2mn + n /*16n*/, *z1mn_up = f + 16 * n,
           *z2mn_up = z1mn_up + n, *z1pl_up = z2mn_up + n, *z2pl_up = z1pl_up + n, *z1mn_dn = z2pl_up + n,
           *z2mn_dn = z1mn_dn + n, *z1pl_dn = z2mn_dn + n, *z2pl_dn = z1pl_dn + n;
#define FF(r, c, i) f[((r) * 4 + (c)) * n + (i)]
    for (int i = 0; i < n; ++i) {
        const double a0 = a[i], a1 = a[n + i], a2 = a[2 * n + i], a3 = a[3 * n + i];
        const double beta = a0 * a1 + 4 * a0 * a3 / 9 + a2 * a3 / 9;                 /* :3388 */
        const double gama = a0 * a1 * a2 * a3 / 9;
        const double l1 = sqrt((beta + sqrt(beta * beta - 4 * gama)) / 2);           /* :3390-3391 */
        const double l2 = sqrt((beta - sqrt(beta * beta - 4 * gama)) / 2);
        o->lam1[i] = l1;
        o->lam2[i] = l2;
        double eta[4] = {0, 0, 0, 0};
        double z1pl = 0, z1mn = 0, z2pl = 0, z2mn = 0;
        if (calculation == 0) {                                                       /* :3397-3416 */
            const double b0 = bb[i], b1 = bb[n + i], b2 = bb[2 * n + i], b3 = bb[3 * n + i];
            const double iu = 1 / ubar0;
            const double x = 1 / ubar0;
            const double Del = 9 * (x * x * x * x - beta * x * x + gama);
            const double D0 = ((a1 * b0 - b1 / ubar0) * (a2 * a3 - 9 / (ubar0 * ubar0)) +
                               2 * (a3 * b2 - 2 * a3 * b0 - 3 * b3 / ubar0) / (ubar0 * ubar0));
            const double D1 = ((a0 * b1 - b0 / ubar0) * (a2 * a3 - 9 / (ubar0 * ubar0)) -
                               2 * a0 * (a3 * b2 - 3 * b3 / ubar0) / ubar0);
            const double D2 = ((a3 * b2 - 3 * b3 / ubar0) * (a0 * a1 - 1 / (ubar0 * ubar0)) -
                               2 * a3 * (a0 * b1 - b0 / ubar0) / ubar0);
            const double D3 = ((a2 * b3 - 3 * b2 / ubar0) * (a0 * a1 - 1 / (ubar0 * ubar0)) +
                               2 * (3 * a0 * b1 - 2 * a0 * b3 - 3 * b0 / ubar0) / (ubar0 * ubar0));
            (void)iu;
            eta[0] = D0 / Del; eta[1] = D1 / Del; eta[2] = D2 / Del; eta[3] = D3 / Del;
            z1pl = (eta[0] / 2 + eta[1] + 5 * eta[2] / 8) * 2 * PI;
            z1mn = (eta[0] / 2 - eta[1] + 5 * eta[2] / 8) * 2 * PI;
            z2pl = (-eta[0] / 8 + 5 * eta[2] / 8 + eta[3]) * 2 * PI;
            z2mn = (-eta[0] / 8 + 5 * eta[2] / 8 - eta[3]) * 2 * PI;
        }
        for (int l = 0; l < 4; ++l) o->eta[l * n + i] = eta[l];
        const double e1 = exp(-clip35(l1 * dtau[i])), e2 = exp(-clip35(l2 * dtau[i]));    /* :3418-3421 */
        const double R1 = -a0 / l1, R2 = -a0 / l2;                                       /* :3423-3425 */
        const double Q1 = 1.0 / 2 * (a0 * a1 / (l1 * l1) - 1), Q2 = 1.0 / 2 * (a0 * a1 / (l2 * l2) - 1);
        const double S1 = -3 / (2 * a3) * (a0 * a1 / l1 - l1), S2 = -3 / (2 * a3) * (a0 * a1 / l2 - l2);
        p1pl[i] = (1.0 / 2 + R1 + 5 * Q1 / 8) * 2 * PI;                                  /* :3427-3434 */
        p2pl[i] = (1.0 / 2 + R2 + 5 * Q2 / 8) * 2 * PI;
        q1pl[i] = (-1.0 / 8 + 5 * Q1 / 8 + S1) * 2 * PI;
        q2pl[i] = (-1.0 / 8 + 5 * Q2 / 8 + S2) * 2 * PI;
        p1mn[i] = (1.0 / 2 - R1 + 5 * Q1 / 8) * 2 * PI;
        p2mn[i] = (1.0 / 2 - R2 + 5 * Q2 / 8) * 2 * PI;
        q1mn[i] = (-1.0 / 8 + 5 * Q1 / 8 - S1) * 2 * PI;
        q2mn[i] = (-1.0 / 8 + 5 * Q2 / 8 - S2) * 2 * PI;
        FF(0, 0, i) = p1mn[i] * e1; FF(0, 1, i) = p1pl[i] / e1; FF(0, 2, i) = p2mn[i] * e2; FF(0, 3, i) = p2pl[i] / e2;   /* :3436-3439 */
        FF(1, 0, i) = q1mn[i] * e1; FF(1, 1, i) = q1pl[i] / e1; FF(1, 2, i) = q2mn[i] * e2; FF(1, 3, i) = q2pl[i] / e2;
        FF(2, 0, i) = p1pl[i] * e1; FF(2, 1, i) = p1mn[i] / e1; FF(2, 2, i) = p2pl[i] * e2; FF(2, 3, i) = p2mn[i] / e2;
        FF(3, 0, i) = q1pl[i] * e1; FF(3, 1, i) = q1mn[i] / e1; FF(3, 2, i) = q2pl[i] * e2; FF(3, 3, i) = q2mn[i] / e2;
        if (calculation == 0) {                                                          /* :3441-3450 */
            const double eu = exp(-clip35(tau[i + 1] / ubar0)), ed = exp(-clip35(tau[i] / ubar0));
            z1mn_up[i] = z1mn * eu; z2mn_up[i] = z2mn * eu; z1pl_up[i] = z1pl * eu; z2pl_up[i] = z2pl * eu;
            z1mn_dn[i] = z1mn * ed; z2mn_dn[i] = z2mn * ed; z1pl_dn[i] = z1pl * ed; z2pl_dn[i] = z2pl * ed;
        } else {                                                                         /* :3451-3459 */
            const double om = 1 - w0[i];
            z1mn_up[i] = om / a0 * (B0[i] / 2 - B1[i] / a1 + B1[i] * dtau[i] / 2) * 2 * PI;
            z2mn_up[i] = -0.5 * om / (4 * a0) * (B0[i] + B1[i] * dtau[i]) * 2 * PI;
            z1pl_up[i] = om / a0 * (B0[i] / 2 + B1[i] / a1 + B1[i] * dtau[i] / 2) * 2 * PI;
            z2pl_up[i] = -0.5 * om / (4 * a0) * (B0[i] + B1[i] * dtau[i]) * 2 * PI;
            z1mn_dn[i] = om / a0 * (B0[i] / 2 - B1[i] / a1) * 2 * PI;
            z2mn_dn[i] = -0.5 * om / (4 * a0) * (B0[i]) * 2 * PI;
            z1pl_dn[i] = om / a0 * (B0[i] / 2 + B1[i] / a1) * 2 * PI;
            z2pl_dn[i] = -0.5 * om / (4 * a0) * (B0[i]) * 2 * PI;
        }
        /* A (:3601-3605): A[j][m] */
        double *A = o->A;
        const double Aj[4][4] = {{1, 1, 1, 1}, {R1, -R1, R2, -R2}, {Q1, Q1, Q2, Q2}, {S1, -S1, S2, -S2}};
        for (int j = 0; j < 4; ++j)
            for (int m = 0; m < 4; ++m) A[(j * 4 + m) * n + i] = Aj[j][m];
    }
    /* matrix fill (:3469-3543) */
    MB(5, 0) = p1mn[0]; MB(5, 1) = q1pl[0]; MB(4, 1) = p1pl[0]; MB(4, 2) = q2mn[0];
    MB(3, 2) = p2mn[0]; MB(3, 3) = q2pl[0]; MB(2, 3) = p2pl[0]; MB(6, 0) = q1mn[0];
    rhs[0] = b_top - z1mn_dn[0];
    rhs[1] = -b_top / 4 - z2mn_dn[0];
    const int nn = n - 1;
    MB(5, N - 2) = FF(2, 2, nn) - rs * FF(0, 2, nn);
    MB(5, N - 1) = FF(3, 3, nn) - rs * FF(1, 3, nn);
    MB(4, N - 1) = FF(2, 3, nn) - rs * FF(0, 3, nn);
    MB(6, N - 3) = FF(2, 1, nn) - rs * FF(0, 1, nn);
    MB(6, N - 2) = FF(3, 2, nn) - rs * FF(1, 2, nn);
    MB(7, N - 4) = FF(2, 0, nn) - rs * FF(0, 0, nn);
    MB(7, N - 3) = FF(3, 1, nn) - rs * FF(1, 1, nn);
    MB(8, N - 4) = FF(3, 0, nn) - rs * FF(1, 0, nn);
    rhs[N - 2] = b_surface - z1pl_up[nn] + rs * z1mn_up[nn];
    rhs[N - 1] = b_surface_SH4 - z2pl_up[nn] + rs * z2mn_up[nn];
    for (int i = 0; i < n - 1; ++i) {
        const int c = 4 * i, k = i + 1;
        MB(5, c + 2) = FF(0, 2, i); MB(5, c + 3) = FF(1, 3, i); MB(5, c + 4) = -p1pl[k]; MB(5, c + 5) = -q1mn[k];
        MB(4, c + 3) = FF(0, 3, i); MB(4, c + 4) = -q1mn[k]; MB(4, c + 5) = -p1mn[k]; MB(4, c + 6) = -q2pl[k];
        MB(3, c + 4) = -p1mn[k]; MB(3, c + 5) = -q1pl[k]; MB(3, c + 6) = -p2pl[k]; MB(3, c + 7) = -q2mn[k];
        MB(2, c + 5) = -p1pl[k]; MB(2, c + 6) = -q2mn[k]; MB(2, c + 7) = -p2mn[k];
        MB(1, c + 6) = -p2mn[k]; MB(1, c + 7) = -q2pl[k];
        MB(0, c + 7) = -p2pl[k];
        MB(6, c + 1) = FF(0, 1, i); MB(6, c + 2) = FF(1, 2, i); MB(6, c + 3) = FF(2, 3, i); MB(6, c + 4) = -q1pl[k];
        MB(7, c) = FF(0, 0, i); MB(7, c + 1) = FF(1, 1, i); MB(7, c + 2) = FF(2, 2, i); MB(7, c + 3) = FF(3, 3, i);
        MB(8, c) = FF(1, 0, i); MB(8, c + 1) = FF(2, 1, i); MB(8, c + 2) = FF(3, 2, i);
        MB(9, c) = FF(2, 0, i); MB(9, c + 1) = FF(3, 1, i);
        MB(10, c) = FF(3, 0, i);
        rhs[c + 2] = z1mn_dn[k] - z1mn_up[i];
        rhs[c + 3] = z2mn_dn[k] - z2mn_up[i];
        rhs[c + 4] = z1pl_dn[k] - z1pl_up[i];
        rhs[c + 5] = z2pl_dn[k] - z2pl_up[i];
    }
    const double fb0 = FF(2, 0, nn), fb1 = FF(2, 1, nn), fb2 = FF(2, 2, nn), fb3 = FF(2, 3, nn);   /* :3546-3550 */
    const double gbot = z1pl_up[nn];
    int info = gb_solve(N, kl, ku, ab, ldab, rhs, ipiv);
    memcpy(o->X, rhs, sizeof(double) * N);
    o->flux_bot = fb0 * rhs[N - 4] + fb1 * rhs[N - 3] + fb2 * rhs[N - 2] + fb3 * rhs[N - 1] + gbot;   /* :2891 */
    if (o->flux) {                                  /* calculate_flux: F.dot(X) + G (:3552-3599, :3631-3635) */
        const double *x = rhs;
        double *fl = o->flux;
        fl[0] = p1mn[0] * x[0] + p1pl[0] * x[1] + p2mn[0] * x[2] + p2pl[0] * x[3] + z1mn_dn[0];
        fl[1] = q1mn[0] * x[0] + q1pl[0] * x[1] + q2mn[0] * x[2] + q2pl[0] * x[3] + z2mn_dn[0];
        fl[2] = p1pl[0] * x[0] + p1mn[0] * x[1] + p2pl[0] * x[2] + p2mn[0] * x[3] + z1pl_dn[0];
        fl[3] = q1pl[0] * x[0] + q1mn[0] * x[1] + q2pl[0] * x[2] + q2mn[0] * x[3] + z2pl_dn[0];
        for (int k = 0; k < n; ++k) {
            const double *xk = x + 4 * k;
            const double g4[4] = {z1mn_up[k], z2mn_up[k], z1pl_up[k], z2pl_up[k]};
            for (int r = 0; r < 4; ++r)
                fl[4 * (k + 1) + r] = FF(r, 0, k) * xk[0] + FF(r, 1, k) * xk[1] + FF(r, 2, k) * xk[2] +
                                      FF(r, 3, k) * xk[3] + g4[r];
        }
    }
    free(p1pl);
    return info;
#undef FF
}

/* One wavelength of setup_2_stream_fluxes + solve (:3189-3333). */
typedef struct {
    double *lam, *q, *eta /*2n*/, *X /*2n*/;
    double flux_bot;
    double *flux;       /* nullable: 2(n+1) moment fluxes F.X + G (fluxes=1, :3311-3331) */
} sh2_out;

static int sh2_column(int n, const double *w0, const double *dtau, const double *tau, const double *a /*2n*/,
                      const double *bb /*2n*/, double b_top, double b_surface, double rs, double ubar0,
                      const double *B0, const double *B1, int calculation, double *ab, int *ipiv,
                      double *rhs, sh2_out *o)
{
    const int kl = 2, ku = 2, ldab = 2 * kl + ku + 1, N = 2 * n;
    memset(ab, 0, sizeof(double) * (size_t)ldab * N);
    memset(rhs, 0, sizeof(double) * N);
    double *Q1 = (double *)malloc(sizeof(double) * n * 12);
    double *Q2 = Q1 + n, *Q1mn = Q2 + n, *Q2mn = Q1mn + n, *Q1pl = Q2mn + n, *Q2pl = Q1pl + n,
           *zmn_up = Q2pl + n, *zpl_up = zmn_up + n, *zmn_dn = zpl_up + n, *zpl_dn = zmn_dn + n;
    for (int i = 0; i < n; ++i) {
        const double a0 = a[i], a1 = a[n + i];
        double eta0 = 0, eta1 = 0;
        if (calculation == 0) {                                         /* :3240-3243 */
            const double Del = ((1 / ubar0) * (1 / ubar0) - a0 * a1);
            eta0 = (bb[n + i] / ubar0 - a1 * bb[i]) / Del;
            eta1 = (bb[i] / ubar0 - a0 * bb[n + i]) / Del;
        }
        o->eta[i] = eta0;
        o->eta[n + i] = eta1;
        const double lam = sqrt(a0 * a1);                               /* :3245-3248 */
        const double e = exp(-clip35(lam * dtau[i]));
        const double q = lam / a1;                                      /* :3251-3256 */
        o->lam[i] = lam;
        o->q[i] = q;
        Q1[i] = (0.5 + q) * 2 * PI;
        Q2[i] = (0.5 - q) * 2 * PI;
        Q1mn[i] = Q1[i] * e; Q2mn[i] = Q2[i] * e; Q1pl[i] = Q1[i] / e; Q2pl[i] = Q2[i] / e;
        if (calculation == 0) {                                         /* :3258-3265 */
            const double zmn = (0.5 * eta0 - eta1) * 2 * PI, zpl = (0.5 * eta0 + eta1) * 2 * PI;
            const double eu = exp(-tau[i + 1] / ubar0), ed = exp(-tau[i] / ubar0);
            zmn_up[i] = zmn * eu; zpl_up[i] = zpl * eu; zmn_dn[i] = zmn * ed; zpl_dn[i] = zpl * ed;
        } else {                                                        /* :3266-3270 */
            const double om = 1 - w0[i];
            zmn_dn[i] = (om / a0 * (B0[i] / 2 - B1[i] / a1)) * 2 * PI;
            zmn_up[i] = (om / a0 * (B0[i] / 2 - B1[i] / a1 + B1[i] * dtau[i] / 2)) * 2 * PI;
            zpl_dn[i] = (om / a0 * (B0[i] / 2 + B1[i] / a1)) * 2 * PI;
            zpl_up[i] = (om / a0 * (B0[i] / 2 + B1[i] / a1 + B1[i] * dtau[i] / 2)) * 2 * PI;
        }
    }
    MB(2, 0) = Q1[0];                                                   /* :3281-3283 */
    MB(1, 1) = Q2[0];
    rhs[0] = b_top - zmn_dn[0];
    const int nn = n - 1;
    MB(3, N - 2) = Q2mn[nn] - rs * Q1mn[nn];                            /* :3287-3289 */
    MB(2, N - 1) = Q1pl[nn] - rs * Q2pl[nn];
    rhs[N - 1] = b_surface - zpl_up[nn] + rs * zmn_up[nn];
    for (int i = 0; i < n - 1; ++i) {                                   /* :3292-3301 */
        const int c = 2 * i, k = i + 1;
        MB(0, c + 3) = -Q2[k];
        MB(1, c + 2) = -Q1[k];
        MB(1, c + 3) = -Q1[k];
        MB(2, c + 1) = Q2pl[i];
        MB(2, c + 2) = -Q2[k];
        MB(3, c) = Q1mn[i];
        MB(3, c + 1) = Q1pl[i];
        MB(4, c) = Q2mn[i];
        rhs[c + 1] = zmn_dn[k] - zmn_up[i];
        rhs[c + 2] = zpl_dn[k] - zpl_up[i];
    }
    const double fb0 = Q2mn[nn], fb1 = Q1pl[nn], gbot = zpl_up[nn];     /* :3305-3309 */
    int info = gb_solve(N, kl, ku, ab, ldab, rhs, ipiv);
    memcpy(o->X, rhs, sizeof(double) * N);
    o->flux_bot = fb0 * rhs[N - 2] + fb1 * rhs[N - 1] + gbot;
    if (o->flux) {                                  /* F.dot(X) + G (:3311-3331) */
        const double *x = rhs;
        double *fl = o->flux;
        fl[0] = Q1[0] * x[0] + Q2[0] * x[1] + zmn_dn[0];
        fl[1] = Q2[0] * x[0] + Q1[0] * x[1] + zpl_dn[0];
        for (int k = 0; k < n; ++k) {
            fl[2 * (k + 1)] = Q1mn[k] * x[2 * k] + Q2pl[k] * x[2 * k + 1] + zmn_up[k];
            fl[2 * (k + 1) + 1] = Q2mn[k] * x[2 * k] + Q1pl[k] * x[2 * k + 1] + zpl_up[k];
        }
    }
    free(Q1);
    return info;
}
#undef MB

#define P(arr, i, w) arr[(size_t)(i) * nwno + (w)]

/* get_reflected_SH (fluxes.py:2675-2976).  f_deltaM is MODIFIED IN PLACE across angles in the TTHG
 * branch exactly as the reference does (:2823-2824): pass a scratch copy. */
int orc_reflected_SH(int nlevel, int nwno, int numg, int numt, const double *dtau, const double *tau,
                     const double *w0, const double *cosb, const double *ftau_cld, const double *ftau_ray,
                     double *f_deltaM, const double *dtau_og, const double *tau_og, const double *w0_og,
                     const double *cosb_og, const double *surf_reflect, const double *ubar0,
                     const double *ubar1, double cos_theta, const double *F0PI, int w_single_form,
                     int w_multi_form, int psingle_form, int w_single_rayleigh, int w_multi_rayleigh,
                     int psingle_rayleigh, double frac_a, double frac_b, double frac_c,
                     double constant_back, double constant_forward, int stream, double b_top,
                     int single_form, double *xint_at_top, double *flux_out /* nullable (numg,numt,stream*nlevel,nwno) */)
{
    const int n = nlevel - 1;
    if (stream != 2 && stream != 4) return 2;
    (void)cosb;
    double *fcol = flux_out ? (double *)malloc(sizeof(double) * stream * (size_t)nlevel) : 0;
    const int N = stream * n, kl = (stream == 4) ? 5 : 2, ldab = 3 * kl + 1;
    double *ab = (double *)malloc(sizeof(double) * ((size_t)ldab * N + 2 * N + 64 * (size_t)n + 64));
    double *rhs = ab + (size_t)ldab * N, *X = rhs + N;
    double *a = X + N, *bb = a + 4 * n, *wsg = bb + 4 * n, *wmu = wsg + 4 * n, *lam1 = wmu + 4 * n,
           *lam2 = lam1 + n, *eta = lam2 + n, *A = eta + 4 * n, *psing = A + 16 * n, *qq = psing + n;
    int *ipiv = (int *)malloc(sizeof(int) * N);
    if (!ab || !ipiv) return 3;
    int rc = 0;
    for (int ig = 0; ig < numg && !rc; ++ig)
        for (int it = 0; it < numt && !rc; ++it) {
            const int fac = ig * numt + it;
            const double u1 = ubar1[fac], u0 = ubar0[fac];
            double Pu0[7], Pu1[7];
            legP(-u0, Pu0);
            legP(u1, Pu1);
            for (int w = 0; w < nwno && !rc; ++w) {
                const double F = F0PI[w], rs = surf_reflect[w];
                for (int i = 0; i < n; ++i) {
                    const double cbo = P(cosb_og, i, w), fc = P(ftau_cld, i, w), fr = P(ftau_ray, i, w);
                    for (int l = 0; l < stream; ++l) { wsg[l * n + i] = 1; wmu[l * n + i] = 1; }
                    double p = 0.0;
                    if (w_single_form == 1 || w_multi_form == 1) {             /* OTHG :2811-2817 */
                        const double fd = P(f_deltaM, i, w);
                        for (int l = 1; l < stream; ++l) {
                            const double ww = (2 * l + 1) * pow(cbo, l);
                            if (w_single_form == 1) wsg[l * n + i] = (ww - (2 * l + 1) * fd) / (1 - fd);
                            if (w_multi_form == 1) wmu[l * n + i] = (ww - (2 * l + 1) * fd) / (1 - fd);
                        }
                    }
                    if (w_single_form == 0 || w_multi_form == 0) {             /* TTHG :2819-2831 */
                        const double gf = constant_forward * cbo, gb = constant_back * cbo;
                        const double f = frac_a + frac_b * pow(gb, frac_c);
                        P(f_deltaM, i, w) *= (f * pow(constant_forward, stream) + (1 - f) * pow(constant_back, stream));
                        const double fd = P(f_deltaM, i, w);
                        for (int l = 1; l < stream; ++l) {
                            const double ww = (2 * l + 1) * (f * pow(gf, l) + (1 - f) * pow(gb, l));
                            if (w_single_form == 0) wsg[l * n + i] = (ww - (2 * l + 1) * fd) / (1 - fd);
                            if (w_multi_form == 0) wmu[l * n + i] = (ww - (2 * l + 1) * fd) / (1 - fd);
                        }
                    }
                    if (w_single_rayleigh == 1) {                              /* :2833-2836 */
                        for (int l = 1; l < stream; ++l) wsg[l * n + i] *= fc;
                        if (stream == 4) wsg[2 * n + i] += 0.5 * fr;
                    }
                    if (w_multi_rayleigh == 1) {                               /* :2837-2840 */
                        for (int l = 1; l < stream; ++l) wmu[l * n + i] *= fc;
                        if (stream == 4) wmu[2 * n + i] += 0.5 * fr;
                    }
                    if (single_form == 0) {                                    /* :2843-2855 */
                        if (psingle_form == 1) {
                            const double s = sqrt(1 + cbo * cbo + 2 * cbo * cos_theta);
                            p = (1 - cbo * cbo) / (s * s * s);
                        } else if (psingle_form == 0) {
                            const double gf = constant_forward * cbo, gb = constant_back * cbo;
                            const double f = frac_a + frac_b * pow(gb, frac_c);
                            const double b1 = 1 + gf * gf + 2 * gf * cos_theta, b2 = 1 + gb * gb + 2 * gb * cos_theta;
                            p = (f * (1 - gf * gf) / sqrt(b1 * b1 * b1) + (1 - f) * (1 - gb * gb) / sqrt(b2 * b2 * b2));
                        }
                        if (psingle_rayleigh == 1) p = fc * p + fr * (0.75 * (1 + cos_theta * cos_theta));
                    }
                    psing[i] = p;
                    for (int l = 0; l < stream; ++l) {                         /* :2858-2860 */
                        a[l * n + i] = (2 * l + 1) - P(w0, i, w) * wmu[l * n + i];
                        bb[l * n + i] = (F * (P(w0, i, w) * wsg[l * n + i])) * Pu0[l] / (4 * PI);
                    }
                }
                const double b_surface = (0. + rs * u0 * F * exp(-P(tau, n, w) / u0));      /* :2863-2865 */
                const double b_surface_SH4 = -(0. + rs * u0 * F * exp(-P(tau, n, w) / u0)) / 4;
                double *dcol = (double *)malloc(sizeof(double) * (3 * (size_t)n + 1));
                double *w0col = dcol + n, *taucol = w0col + n;
                for (int i = 0; i < n; ++i) { dcol[i] = P(dtau, i, w); w0col[i] = P(w0, i, w); }
                for (int i = 0; i <= n; ++i) taucol[i] = P(tau, i, w);
                double flux_bot;
                if (stream == 2) {
                    sh2_out o = {lam1, qq, eta, X, 0, fcol};
                    rc = sh2_column(n, w0col, dcol, taucol, a, bb, b_top, b_surface, rs, u0, 0, 0, 0, ab, ipiv, rhs, &o);
                    flux_bot = o.flux_bot;
                } else {
                    sh4_out o = {n, lam1, lam2, eta, A, X, 0, fcol};
                    rc = sh4_column(n, w0col, dcol, taucol, a, bb, b_top, b_surface, b_surface_SH4, rs, u0, 0, 0, 0,
                                    ab, ipiv, rhs, &o);
                    flux_bot = o.flux_bot;
                }
                free(dcol);
                if (rc) break;
                if (fcol)
                    for (int r = 0; r < stream * nlevel; ++r)
                        flux_out[((size_t)fac * stream * nlevel + r) * nwno + w] = fcol[r];
                const double mus = (u1 + u0) / (u1 * u0);                                    /* :2900 */
                double xint = flux_bot / PI;                                                /* :2967 */
                /* the recursion runs bottom-up; integrals are per layer */
                for (int i = n - 1; i >= 0; --i) {
                    const double dt = P(dtau, i, w), w0_ = P(w0, i, w);
                    const double exptrm_mus = (1 - exp(-clip35(mus * dt))) / mus;             /* :2901-2902 */
                    const double exptau_mu = exp(-clip35(P(tau, i, w) * 1 / u0));             /* :2903-2904 */
                    const double expon1 = exptrm_mus * exptau_mu;
                    double multi;
                    if (stream == 2) {                                                       /* :2907-2922 */
                        const double alpha = 1 / u1 + lam1[i], beta = 1 / u1 - lam1[i];
                        const double ea = (1 - exp(-clip35(alpha * dt))) / alpha;
                        const double eb = (1 - exp(-clip35(beta * dt))) / beta;
                        const double A0 = X[2 * i] * (wmu[i] - wmu[n + i] * Pu1[1] * qq[i]) * ea;
                        const double A1 = X[2 * i + 1] * (wmu[i] + wmu[n + i] * Pu1[1] * qq[i]) * eb;
                        const double N0 = wmu[i] * (eta[i] * expon1);
                        const double N1 = wmu[n + i] * Pu1[1] * (eta[n + i] * expon1);
                        multi = A0 + N0 + A1 + N1;
                    } else {                                                                 /* :2924-2952 */
                        const double al1 = 1 / u1 + lam1[i], al2 = 1 / u1 + lam2[i], be1 = 1 / u1 - lam1[i],
                                     be2 = 1 / u1 - lam2[i];
                        double ex[4];
                        ex[0] = (1 - exp(-clip35(al1 * dt))) / al1 * X[4 * i];
                        ex[1] = (1 - exp(-clip35(be1 * dt))) / be1 * X[4 * i + 1];
                        ex[2] = (1 - exp(-clip35(al2 * dt))) / al2 * X[4 * i + 2];
                        ex[3] = (1 - exp(-clip35(be2 * dt))) / be2 * X[4 * i + 3];
                        double Aint[4];
                        for (int m = 0; m < 4; ++m) {
                            double s = 0;
                            for (int j = 0; j < 4; ++j) s = s + wmu[j * n + i] * Pu1[j] * A[(j * 4 + m) * n + i];
                            Aint[m] = s * ex[m];
                        }
                        const double N0 = wmu[i] * Pu1[0] * eta[i] * expon1, N1 = wmu[n + i] * Pu1[1] * eta[n + i] * expon1,
                                     N2 = wmu[2 * n + i] * Pu1[2] * eta[2 * n + i] * expon1,
                                     N3 = wmu[3 * n + i] * Pu1[3] * eta[3 * n + i] * expon1;
                        multi = (Aint[0] + N0 + Aint[1] + N1 + Aint[2] + N2 + Aint[3] + N3);
                    }
                    double ps = psing[i];
                    if (single_form == 1)                                                     /* :2954-2957 */
                        for (int l = 0; l < stream; ++l) ps = ps + wsg[l * n + i] * Pu0[l] * Pu1[l];
                    const double e1 = exp(-clip35(mus * P(dtau_og, i, w)));                   /* :2959-2960 */
                    const double intg = (w0_ * multi + P(w0_og, i, w) * F / (4 * PI) * ps * (1 - e1) *
                                                            exp(-P(tau_og, i, w) / u0) / mus);  /* :2961-2965 */
                    xint = (xint * exp(-dt / u1) + intg / u1);                                 /* :2968-2970 */
                }
                xint_at_top[(size_t)fac * nwno + w] = xint;
            }
        }
    free(ab);
    free(ipiv);
    free(fcol);
    return rc;
}

static double planck_lambda(double t, double wcm)   /* fluxes.py:1660-1680 */
{
    const double h = 6.62607004e-27, c = 2.99792458e+10, k = 1.38064852e-16;
    return ((2.0 * h * (c * c)) / pow(wcm, 5.0)) * (1.0 / (exp((h * c) / (t * (wcm * k))) - 1.0));
}

/* get_thermal_SH (fluxes.py:2979-3186), flx = 0. */
int orc_thermal_SH(int nlevel, const double *wno, int nwno, int numg, int numt, const double *tlevel,
                   const double *dtau, const double *tau, const double *w0, const double *cosb,
                   const double *cosb_og, const double *plevel, const double *ubar1,
                   const double *surf_reflect, int stream, int hard_surface, double *xint_at_top)
{
    const int n = nlevel - 1;
    if (stream != 2 && stream != 4) return 2;
    const double mu1 = 0.5;
    const int N = stream * n, kl = (stream == 4) ? 5 : 2, ldab = 3 * kl + 1;
    double *ab = (double *)malloc(sizeof(double) * ((size_t)ldab * N + 2 * N + 64 * (size_t)n + (size_t)nlevel + 64));
    double *rhs = ab + (size_t)ldab * N, *X = rhs + N;
    double *a = X + N, *bb = a + 4 * n, *wmu = bb + 4 * n, *lam1 = wmu + 4 * n, *lam2 = lam1 + n,
           *eta = lam2 + n, *A = eta + 4 * n, *qq = A + 16 * n, *b0 = qq + n, *b1 = b0 + n, *allb = b1 + n,
           *dcol = allb + nlevel, *w0col = dcol + n, *taucol = w0col + n;
    int *ipiv = (int *)malloc(sizeof(int) * N);
    /* ff = 0 if cosb == cosb_og everywhere else cosb_og**stream (np.array_equal, :3072-3075) */
    int same = 1;
    for (size_t k = 0; k < (size_t)n * nwno; ++k)
        if (cosb[k] != cosb_og[k]) { same = 0; break; }
    int rc = 0;
    for (int w = 0; w < nwno && !rc; ++w) {
        const double rs = surf_reflect[w];
        for (int l = 0; l < nlevel; ++l) allb[l] = planck_lambda(tlevel[l], 1.0 / wno[w]);   /* :3058 */
        for (int i = 0; i < n; ++i) {
            dcol[i] = P(dtau, i, w);
            w0col[i] = P(w0, i, w);
            b0[i] = allb[i];
            b1[i] = (allb[i + 1] - b0[i]) / dcol[i];                                          /* :3060 */
            const double cbo = P(cosb_og, i, w);
            const double ff = same ? 0. * cbo : pow(cbo, stream);
            for (int l = 0; l < stream; ++l) {                                                /* :3081-3083 */
                wmu[l * n + i] = (2 * l + 1) * (pow(cbo, l) - ff) / (1 - ff);
                a[l * n + i] = (2 * l + 1) - w0col[i] * wmu[l * n + i];
                bb[l * n + i] = 0.0;
            }
        }
        for (int i = 0; i <= n; ++i) taucol[i] = P(tau, i, w);
        const double tau_top = dcol[0] * plevel[0] / (plevel[1] - plevel[0]);                /* :3062 */
        const double b_top = PI * (1.0 - exp(-tau_top / mu1)) * allb[0];                      /* :3063 */
        const double b_surface = hard_surface ? PI * allb[nlevel - 1]                         /* :3065-3068 */
                                              : PI * (allb[nlevel - 1] + b1[n - 1] * mu1);
        const double b_surface_SH4 = (-PI * allb[nlevel - 1] / 4);                            /* :3070 */
        double flux_bot;
        if (stream == 2) {
            sh2_out o = {lam1, qq, eta, X, 0};
            rc = sh2_column(n, w0col, dcol, taucol, a, bb, b_top, b_surface, rs, 0, b0, b1, 1, ab, ipiv, rhs, &o);
            flux_bot = o.flux_bot;
        } else {
            sh4_out o = {n, lam1, lam2, eta, A, X, 0};
            rc = sh4_column(n, w0col, dcol, taucol, a, bb, b_top, b_surface, b_surface_SH4, rs, 0, b0, b1, 1, ab,
                            ipiv, rhs, &o);
            flux_bot = o.flux_bot;
        }
        (void)flux_bot;
        if (rc) break;
        for (int fac = 0; fac < numg * numt; ++fac) {
            const double u1 = ubar1[fac];
            double Pu1[7];
            legP(u1, Pu1);
            double xint = hard_surface ? allb[nlevel - 1] * 2 * PI                            /* :3173-3176 */
                                       : (allb[nlevel - 1] + b1[n - 1] * u1) * 2 * PI;
            for (int i = n - 1; i >= 0; --i) {
                const double dt = dcol[i], w0_ = w0col[i];
                double multi;
                if (stream == 2) {                                                            /* :3116-3131 */
                    const double alpha = 1 / u1 + lam1[i], beta = 1 / u1 - lam1[i];
                    const double ea = (1 - exp(-clip35(alpha * dt))) / alpha, eb = (1 - exp(-clip35(beta * dt))) / beta;
                    const double A0 = X[2 * i] * (wmu[i] - wmu[n + i] * Pu1[1] * qq[i]) * ea;
                    const double A1 = X[2 * i + 1] * (wmu[i] + wmu[n + i] * Pu1[1] * qq[i]) * eb;
                    const double ed = exp(-dt / u1);
                    const double N0 = wmu[i] * ((1 - w0_) * u1 / a[i] * (b0[i] * (1 - ed) + b1[i] * (u1 - (dt + u1) * ed)));
                    const double N1 = wmu[n + i] * Pu1[1] * ((1 - w0_) * u1 / a[i] * (b1[i] * (1 - ed) / a[n + i]));
                    multi = A0 + N0 + A1 + N1;
                } else {                                                                      /* :3133-3160 */
                    const double al1 = 1 / u1 + lam1[i], al2 = 1 / u1 + lam2[i], be1 = 1 / u1 - lam1[i],
                                 be2 = 1 / u1 - lam2[i];
                    double ex[4];
                    ex[0] = (1 - exp(-clip35(al1 * dt))) / al1 * X[4 * i];
                    ex[1] = (1 - exp(-clip35(be1 * dt))) / be1 * X[4 * i + 1];
                    ex[2] = (1 - exp(-clip35(al2 * dt))) / al2 * X[4 * i + 2];
                    ex[3] = (1 - exp(-clip35(be2 * dt))) / be2 * X[4 * i + 3];
                    double Aint[4];
                    for (int m = 0; m < 4; ++m) {
                        double s = 0;
                        for (int j = 0; j < 4; ++j) s = s + wmu[j * n + i] * Pu1[j] * A[(j * 4 + m) * n + i];
                        Aint[m] = s * ex[m];
                    }
                    const double ed = exp(-clip35(dt / u1));                                  /* :3154 */
                    const double N0 = wmu[i] * ((1 - w0_) * u1 / a[i] * (b0[i] * (1 - ed) + b1[i] * (u1 - (dt + u1) * ed)));
                    const double N1 = wmu[n + i] * u1 * ((1 - w0_) * u1 / a[i] * (b1[i] * (1 - ed) / a[n + i]));
                    multi = Aint[0] + Aint[1] + Aint[2] + Aint[3] + N0 + N1 + 0.0 + 0.0;
                }
                const double ed = exp(-(dt / u1));                                            /* :3163-3165 */
                const double intg = (w0_ * multi * 2 * PI +
                                     2 * PI * (1 - w0_) * u1 * (b0[i] * (1 - ed) + b1[i] * (u1 - (dt + u1) * ed)));
                xint = (xint * exp(-dt / u1) + intg / u1);                                    /* :3178-3180 */
            }
            xint_at_top[(size_t)fac * nwno + w] = xint;
        }
    }
    free(ab);
    free(ipiv);
    return rc;
}
