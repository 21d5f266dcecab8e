// Spherical-harmonics (SH2 / SH4) reflected light and thermal emission -- gfx950.
//
// Replaces fluxes.get_reflected_SH / get_thermal_SH with setup_2_stream_fluxes,
// setup_4_stream_fluxes and solve_4_stream_banded (reference picaso/fluxes.py:2675-3628).  The
// reference assembles a banded matrix (5 diagonals for SH2, 11 for SH4: 3.2 GB at 1e5 x 90) per
// angle and calls LAPACK dgbsv per wavelength.  Here nothing is assembled: per layer the stream
// coefficients split into NB = stream/2 decaying-mode unknowns d and NB growing-mode unknowns
// u = E v (E = diag exp(-lambda_m dtau), so v is the bounded value at the layer bottom); with the
// NB x NB blocks Mn, Pl of the reference's boundary rows the moment fluxes are
//     top :  Fmn = Mn d + Pl E v + zmn_dn      Fpl = Pl d + Mn E v + zpl_dn
//     bot :  Fmn = Mn E d + Pl v + zmn_up      Fpl = Pl E d + Mn v + zpl_up
// (SH4: Mn = [[p1mn,p2mn],[q1mn,q2mn]], Pl = [[p1pl,p2pl],[q1pl,q2pl]], fluxes.py:3427-3543;
//  SH2: Mn = Q1, Pl = Q2, fluxes.py:3251-3301).  One top-down sweep carries the relation
// d_i = delta_i - R_i v_i that everything above imposes on layer i and the TOA functional
// J = kappa + zeta.v_i of the source-function integrals (fluxes.py:2898-2970, 3105-3182); the
// surface rows fix v_{n-1}.  Only decaying exponentials appear, the blocks are the physical
// reflection operators, and no pivoting across layers is needed: tools/sh_sweep_numpy.py agrees
// with the reference's pivoted LAPACK solve to 2e-13 on thin, thick (35-clipped) and conservative
// columns (tests/test_single_sweep_numpy.py).  One lane per wavelength, every plane read once per
// angle, coalesced in the reference's (nlayer, nwno) layout.
//
// Reference quirk kept: in the TTHG branch f_deltaM is multiplied in place once per angle
// (fluxes.py:2823-2824), so angle k (in (g,t) order) sees f_deltaM * fac^(k+1).
#include <type_traits>

#include "common.hpp"
#include "device_math.hpp"

namespace pz {

constexpr int SH_MAX_ANG = 16;
// A/B switches of the round-2 instruction-count work, both measured slower and left off
// (AB_CMD="python tools/sh_time.py" tools/ab.sh, SH4 1e5 x 90 x 5, steady state, one box: 1.131 ms with
// both off):
//   PZ_SH_OPT_EXP  exp(-tau[i+1]/u0) and exp(-tau_og/u0) as products of exponentials already at hand
//                  when the wave's planes allow it (two of five exponentials per layer): 1.305 ms --
//                  the wave-uniform tests and branches cost more than the polynomials they skip;
//   PZ_SH_OPT_NC   odd-moment terms of eta / cm / Nsum skipped on cloud-free layers: 1.165 ms.
#ifndef PZ_SH_OPT_EXP
#define PZ_SH_OPT_EXP 0
#endif
#ifndef PZ_SH_OPT_NC
#define PZ_SH_OPT_NC 0
#endif

struct SHArgs {
    int nlayer, nwno, stream;
    long pitch;
    const double *dtau, *tau, *w0, *ftau_cld, *ftau_ray, *f_deltaM, *dtau_og, *tau_og, *w0_og, *cosb_og;
    const double *surf_reflect, *F0PI;
    // angles of this launch chunk; constants derived on the host (wave-uniform -> SGPR)
    struct Angle {
        double u0, u1, iu0, iu1, mus, imus, nl0, nl1, nlm;   // nl* = -log2(e) {1/u0, 1/u1, mus}
    } ang[SH_MAX_ANG];
    int first_angle, compound, nang;                         // reference angle index of ang[0]; angles of this launch
    int xcd_order;                                           // block order, see k_sh
    unsigned ncg;                                            // column groups (blocks per angle)
    double cos_theta;
    int w_single_form, w_multi_form, psingle_form, w_single_rayleigh, w_multi_rayleigh,
        psingle_rayleigh, single_form;
    double frac_a, frac_b, frac_c, constant_back, constant_forward, b_top;
    // thermal
    const double *wno, *tlevel, *plevel;     // device (nwno) / (nlevel) / (nlevel)
    int hard_surface, use_ff;                // use_ff: cosb != cosb_og somewhere (fluxes.py:3072-3075)
    double *xint;                            // (angles of this launch, nwno)
    // flx = 1 (reflected): layer moment fluxes F.X + G (calculate_flux, fluxes.py:3631-3635)
    double *flux;                            // (angles of this launch, stream*nlevel, nwno)
    double *scratch;                         // (angles, 4 NB^2 + 5 NB, nlayer, nwno) sweep-1 state
    // batched launch (picaso_get_reflected_SH_batch_dev): blockIdx.y = spectrum, see ReflectedArgs::batch
    const struct SHBatchItem *batch;
    int nspec_;
    // resume below a cloud-free top (picaso_get_reflected_SH_top_dev): the sweep state k_sh4_clear left at the top of
    // layer `start_layer`, [slot][nwno] (SHC_SHARED shared slots, then SHC_STATE per angle of the call)
    int start_layer;
    const double *state;
};
constexpr int SHC_SHARED = 20, SHC_STATE = 11;
struct SHBatchItem {
    const double *dtau, *tau, *w0, *ftau_cld, *ftau_ray, *f_deltaM, *dtau_og, *tau_og, *w0_og, *cosb_og;
    const double *surf_reflect, *F0PI;
    double *xint;
    double cos_theta;
    SHArgs::Angle ang[SH_MAX_ANG];
};

__device__ __forceinline__ double clip35(double x) { return fmin(fmax(x, -35.0), 35.0); }   // slice_rav
constexpr double LOG2E = LOG2E_D;
constexpr double EXP_M35 = 6.305116760146989e-16;        // exp(-35)
// exp(-clip35(y/u)) for y >= 0 given t = y * (-log2(e)/u): only the lower clip can bind
__device__ __forceinline__ double fexp2_clip(double t, const Exp2Coef &K) { return fexp2(fmax(t, -35.0 * LOG2E), K); }

// x / d for a compile-time divisor, correctly rounded like the division it replaces, in 3 instructions
// instead of the 11 of v_div_scale / v_rcp / v_div_fmas / v_div_fixup: q = x RN(1/d), r = x - d q (exact in
// the fma), q' = q + r RN(1/d) (Markstein's quotient refinement; checked against x/d on 2e5 values each for
// d = 9 and d = 4 pi).  Four such divisions per layer (three /9 in the SH4 quartic, one /(4 pi) in the
// single-scattering term, fluxes.py:3388-3391, :2959) were 7 % of the SH4 kernel.
__device__ __forceinline__ double div_const(double x, double d, double rd)
{
#pragma clang fp contract(off)
    const double q = x * rd;
    const double r = fma(-d, q, x);
    return fma(r, rd, q);
}
constexpr double R9 = 1.0 / 9.0, FOURPI = 4 * PI, R4PI = 1.0 / (4 * PI);

template <int NB>
struct Blk {
    double m[NB][NB];
};

template <int NB>
__device__ __forceinline__ Blk<NB> mm(const Blk<NB> &A, const Blk<NB> &B)
{
    Blk<NB> C;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < NB; ++k) s += A.m[i][k] * B.m[k][j];
            C.m[i][j] = s;
        }
    return C;
}

template <int NB>
__device__ __forceinline__ void mv(const Blk<NB> &A, const double (&x)[NB], double (&y)[NB])
{
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < NB; ++k) s += A.m[i][k] * x[k];
        y[i] = s;
    }
}

template <int NB>
__device__ __forceinline__ void mtv(const Blk<NB> &A, const double (&x)[NB], double (&y)[NB])   // A^T x
{
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < NB; ++k) s += A.m[k][i] * x[k];
        y[i] = s;
    }
}

template <int NB>
__device__ __forceinline__ Blk<NB> inv(const Blk<NB> &A)
{
    Blk<NB> R;
    if (NB == 1) {
        R.m[0][0] = frcp(A.m[0][0]);
    } else {
        const double idet = frcp(A.m[0][0] * A.m[NB - 1][NB - 1] - A.m[0][NB - 1] * A.m[NB - 1][0]);
        R.m[0][0] = A.m[NB - 1][NB - 1] * idet;
        R.m[NB - 1][NB - 1] = A.m[0][0] * idet;
        R.m[0][NB - 1] = -A.m[0][NB - 1] * idet;
        R.m[NB - 1][0] = -A.m[NB - 1][0] * idet;
    }
    return R;
}

template <int NB>
__device__ __forceinline__ Blk<NB> scale_cols(const Blk<NB> &A, const double (&e)[NB])   // A diag(e)
{
    Blk<NB> R;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) R.m[i][j] = A.m[i][j] * e[j];
    return R;
}

template <int NB>
__device__ __forceinline__ double dot(const double (&a)[NB], const double (&b)[NB])
{
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NB; ++i) s += a[i] * b[i];
    return s;
}

__device__ __forceinline__ void legP4(double mu, double (&P)[4])   // fluxes.py:3639-3646, l = 0..3
{
    P[0] = 1;
    P[1] = mu;
    P[2] = (3 * mu * mu - 1) / 2;
    P[3] = (5 * mu * mu * mu - 3 * mu) / 2;
}

// Per-layer mode structure from the a_l coefficients.
template <int NB>
struct Modes {
    double lam[NB], E[NB];
    Blk<NB> Mn, Pl;
    double cA[2 * NB][2 * NB];   // A[j][m] of fluxes.py:3601-3605 (SH4); unused for SH2
    double q;                    // SH2: lam/a1
    double beta, gama;           // SH4 quartic coefficients (particular solution)
};

__device__ __forceinline__ void modes_sh4(const double (&a)[4], double dt, Modes<2> &M, const Exp2Coef &K)
{
    const double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
    M.beta = a0 * a1 + div_const(4 * a0 * a3, 9.0, R9) + div_const(a2 * a3, 9.0, R9);   // fluxes.py:3388-3391
    M.gama = div_const(a0 * a1 * a2 * a3, 9.0, R9);
    const double disc = fsqrt(M.beta * M.beta - 4 * M.gama);
    // lam = x rsqrt(x) and 1/lam = rsqrt(x) from one v_rsq_f64 + Newton (frsq, ~1 ulp) instead of a
    // correctly rounded sqrt followed by a reciprocal (fluxes.py:3393-3394, 3423-3425)
    const double x1 = (M.beta + disc) / 2, x2 = (M.beta - disc) / 2;
    const double il1 = frsq(x1), il2 = frsq(x2);
    const double l1 = x1 * il1, l2 = x2 * il2;
    M.lam[0] = l1;
    M.lam[1] = l2;
    const double a01 = a0 * a1, s3 = -1.5 * frcp(a3);
    const double R1 = -a0 * il1, R2 = -a0 * il2;                      // :3423-3425
    const double Q1 = 0.5 * (a01 * il1 * il1 - 1), Q2 = 0.5 * (a01 * il2 * il2 - 1);
    const double S1 = s3 * (a01 * il1 - l1), S2 = s3 * (a01 * il2 - l2);
    const double tp = 2 * PI;
    M.Pl.m[0][0] = (0.5 + R1 + 5 * Q1 / 8) * tp;                      // p1pl  :3427-3434
    M.Pl.m[0][1] = (0.5 + R2 + 5 * Q2 / 8) * tp;                      // p2pl
    M.Pl.m[1][0] = (-0.125 + 5 * Q1 / 8 + S1) * tp;                   // q1pl
    M.Pl.m[1][1] = (-0.125 + 5 * Q2 / 8 + S2) * tp;                   // q2pl
    M.Mn.m[0][0] = (0.5 - R1 + 5 * Q1 / 8) * tp;                      // p1mn
    M.Mn.m[0][1] = (0.5 - R2 + 5 * Q2 / 8) * tp;                      // p2mn
    M.Mn.m[1][0] = (-0.125 + 5 * Q1 / 8 - S1) * tp;                   // q1mn
    M.Mn.m[1][1] = (-0.125 + 5 * Q2 / 8 - S2) * tp;                   // q2mn
    M.E[0] = fexpk(-clip35(l1 * dt), K);                              // :3418-3421
    M.E[1] = fexpk(-clip35(l2 * dt), K);
    // A[j][m], m = (d0, u0, d1, u1)
    const double Aj[4][4] = {{1, 1, 1, 1}, {R1, -R1, R2, -R2}, {Q1, Q1, Q2, Q2}, {S1, -S1, S2, -S2}};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < 4; ++m) M.cA[j][m] = Aj[j][m];
}

// modes_sh4 for reflected light with the layer's six independent reciprocals -- 1/a3, 1/Delta(1/u0) of the particular
// solution (:3397), 1/((1/u1)^2 - lam_r^2) and 1/E_r of the source-function weights (:2929-2937) -- taken from ONE
// v_rcp_f64 + Newton core (Montgomery's trick: prefix products, one reciprocal, 2 multiplies per value on the way back;
// 23 instructions instead of 6 x 5, and five fewer quarter-rate v_rcp_f64).  The six values of a layer span
// 1e-31 (E at the 35 clip, twice) .. 1e9 (Delta at grazing incidence): their product stays far inside the fp64 range.
#ifndef PZ_SH_BATCH_RCP
#define PZ_SH_BATCH_RCP 1
#endif
struct ShRcp { double iDel, rab[2], iE[2]; };
__device__ __forceinline__ void modes_sh4_batched(const double (&a)[4], double dt, double iu0, double iu1, Modes<2> &M,
                                                  ShRcp &Q, const Exp2Coef &K)
{
    const double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
    M.beta = a0 * a1 + div_const(4 * a0 * a3, 9.0, R9) + div_const(a2 * a3, 9.0, R9);   // fluxes.py:3388-3391
    M.gama = div_const(a0 * a1 * a2 * a3, 9.0, R9);
    const double disc = fsqrt(M.beta * M.beta - 4 * M.gama);
    const double x1 = (M.beta + disc) / 2, x2 = (M.beta - disc) / 2;
    const double il1 = frsq(x1), il2 = frsq(x2);
    const double l1 = x1 * il1, l2 = x2 * il2;
    M.lam[0] = l1;
    M.lam[1] = l2;
    M.E[0] = fexpk(-clip35(l1 * dt), K);                              // :3418-3421
    M.E[1] = fexpk(-clip35(l2 * dt), K);
    {
        const double xx = iu0 * iu0;
        const double v0 = a3, v1 = 9 * (xx * xx - M.beta * xx + M.gama), v2 = (iu1 + l1) * (iu1 - l1),
                     v3 = (iu1 + l2) * (iu1 - l2), v4 = M.E[0], v5 = M.E[1];
        const double p1 = v0 * v1, p2 = p1 * v2, p3 = p2 * v3, p4 = p3 * v4, p5 = p4 * v5;
        double r = frcp(p5);
        Q.iE[1] = r * p4;
        r *= v5;
        Q.iE[0] = r * p3;
        r *= v4;
        Q.rab[1] = r * p2;
        r *= v3;
        Q.rab[0] = r * p1;
        r *= v2;
        Q.iDel = r * v0;
        r *= v1;                                                      // 1/a3
        const double a01 = a0 * a1, s3 = -1.5 * r;
        const double R1 = -a0 * il1, R2 = -a0 * il2;                  // :3423-3425
        const double Q1 = 0.5 * (a01 * il1 * il1 - 1), Q2 = 0.5 * (a01 * il2 * il2 - 1);
        const double S1 = s3 * (a01 * il1 - l1), S2 = s3 * (a01 * il2 - l2);
        const double tp = 2 * PI;
        M.Pl.m[0][0] = (0.5 + R1 + 5 * Q1 / 8) * tp;                  // :3427-3434
        M.Pl.m[0][1] = (0.5 + R2 + 5 * Q2 / 8) * tp;
        M.Pl.m[1][0] = (-0.125 + 5 * Q1 / 8 + S1) * tp;
        M.Pl.m[1][1] = (-0.125 + 5 * Q2 / 8 + S2) * tp;
        M.Mn.m[0][0] = (0.5 - R1 + 5 * Q1 / 8) * tp;
        M.Mn.m[0][1] = (0.5 - R2 + 5 * Q2 / 8) * tp;
        M.Mn.m[1][0] = (-0.125 + 5 * Q1 / 8 - S1) * tp;
        M.Mn.m[1][1] = (-0.125 + 5 * Q2 / 8 - S2) * tp;
        const double Aj[4][4] = {{1, 1, 1, 1}, {R1, -R1, R2, -R2}, {Q1, Q1, Q2, Q2}, {S1, -S1, S2, -S2}};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m) M.cA[j][m] = Aj[j][m];
    }
}

__device__ __forceinline__ void modes_sh2(const double (&a)[2], double dt, Modes<1> &M, const Exp2Coef &K)
{
    const double lam = sqrt(a[0] * a[1]);                             // fluxes.py:3245
    M.lam[0] = lam;
    M.q = lam * frcp(a[1]);                                           // :3251
    M.Mn.m[0][0] = (0.5 + M.q) * 2 * PI;                              // Q1
    M.Pl.m[0][0] = (0.5 - M.q) * 2 * PI;                              // Q2
    M.E[0] = fexpk(-clip35(lam * dt), K);                             // :3246-3248
}

// One kernel for stream 2 / 4 (NB = 1 / 2), reflected / thermal.
// FAST: the reference's default SH options (config.json: TTHG weights for single and multiple
// scattering, TTHG single-scattering phase function, Rayleigh in all three, explicit single form,
// frac_c = 2) fixed at compile time: the nine option words are branches inside the layer loop otherwise.
#ifndef PZ_SH_MINWAVES
#define PZ_SH_MINWAVES 2
#endif
// DRVT (FAST reflected kernels): the level planes tau and tau_og are not in HBM (NULL).  compute_opacity forms them as
// running sums of dtau / dtau_og from 0 at the top (optics.py:353-354, 418-420), so the beam exponentials of a level,
// exp(-tau/u0) and exp(-tau_og/u0), are running PRODUCTS of the layers' exp(-dtau/u0): carried unclipped down the column,
// one multiply per layer -- in the symmetric geometry (u0 == u1) by exp(-dtau/u1), which the layer needs anyway, so
// two of its five exponentials (15 instructions each) and three of its eleven loads go; other geometries pay one
// exponential for the two.  The reference's clipped form exp(-clip35(tau/u0)) (:3418-3421) is max(., e^-35) of the
// unclipped one.  A column's value differs from the plane-reading kernel's by the rounding of the product (<= n ulp).
template <int NB, bool THERMAL, bool FLX, bool FAST, bool DRVT, typename AnglePtr>
__device__ __forceinline__ void sh_body(const SHArgs &a, AnglePtr angp)
{
    static_assert(!DRVT || (FAST && !THERMAL && !FLX), "derived level planes: the default-options reflected kernels");
    constexpr int NS = 2 * NB;      // stream
    // 1-D grid, XCD-aware order.  Consecutive workgroups go to consecutive XCDs (8 of them, each with
    // its own L2), and every angle re-reads the same 13 planes: block b = (chunk of 8 column groups,
    // angle, column group within the chunk), so the nang blocks of one column group are dispatched
    // within 8*nang blocks of each other, all on XCD (column group % 8): the first one's plane rows are
    // L2 hits for the others (HBM fetch 3.6 GB -> ~1.2 GB per 1e5 x 90 x 5 spectrum, PMC).
    // Small launches (about one block per CU or less) keep the plain angle-major order: there the XCD
    // grouping would put five blocks of a column group on one XCD while others stay short of work.
    const int nang = a.nang;
    const unsigned b = blockIdx.x;
    int ang;
    long w;
    if (a.xcd_order) {
        const unsigned chunk = b / (8u * nang), rem = b - chunk * (8u * nang);
        ang = (int)(rem >> 3);
        w = ((long)chunk * 8 + (rem & 7u)) * blockDim.x + threadIdx.x;
    } else {
        ang = (int)(b / a.ncg);
        w = (long)(b - (unsigned)ang * a.ncg) * blockDim.x + threadIdx.x;
    }
    if (w >= a.nwno) return;
    const int n = a.nlayer;
    const long pitch = a.pitch;
    // Plane element of a layer (`LD` in the layer body).  FAST kernels address the planes as SGPR base + 32-bit lane
    // offset (launch_sh takes them only for planes smaller than 4 GB): one v_add_u32 per layer for all planes instead of a
    // 64-bit v_lshl_add_u64 per load (eleven per layer; profiles/r05_isa_k_sh4_fast_before.json).
    const unsigned voff0 = (unsigned)(w * 8), pitch8 = (unsigned)(pitch * 8);
    const int w_single_form = FAST ? 0 : a.w_single_form, w_multi_form = FAST ? 0 : a.w_multi_form;
    const int psingle_form = FAST ? 0 : a.psingle_form, single_form = FAST ? 0 : a.single_form;
    const int w_single_rayleigh = FAST ? 1 : a.w_single_rayleigh, w_multi_rayleigh = FAST ? 1 : a.w_multi_rayleigh;
    const int psingle_rayleigh = FAST ? 1 : a.psingle_rayleigh;
    const double frac_c = FAST ? 2.0 : a.frac_c;
    const auto &g = angp[ang];
    const double u0 = g.u0, u1 = g.u1, ct = a.cos_theta;
    const int fd_power = a.compound ? a.first_angle + ang + 1 : 1;   // compounded f_deltaM
    double *const xint = a.xint + (long)ang * a.nwno;
    const double F = THERMAL ? 0.0 : a.F0PI[w], rs = a.surf_reflect[w];
    double Pu0[4], Pu1[4];
    legP4(-u0, Pu0);
    legP4(u1, Pu1);
    const double mus = g.mus, iu1 = g.iu1, iu0 = g.iu0, imus = g.imus;
    const bool sym = (u0 == u1);                                     // mus = 2/u1
    Exp2Coef K;
    K.load();

    // thermal: Planck at the levels (fluxes.py:3058-3060)
    const double wn = THERMAL ? a.wno[w] : 0.0;
    double Bn = THERMAL ? planck_lambda(a.tlevel[0], wn, K) : 0.0;
    const double B_top = Bn;
    double b1_last = 0.0;

    double T = 1.0, kappa = 0.0;
    double e_top = 0.0;              // exp(-tau/u0) at the top of the current layer (carried)
    double p_og = 1.0;               // DRVT: exp(-tau_og/u0) at the top of the current layer, unclipped (e_top: exp(-tau/u0))
    if (DRVT) e_top = 1.0;           // tau[0] = 0
    double zeta[NB], delta[NB];
    Blk<NB> R;
    // previous layer
    Blk<NB> pMn, pPl, pME, pPE;
    double p_zmn_up[NB], p_zpl_up[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) { zeta[i] = delta[i] = p_zmn_up[i] = p_zpl_up[i] = 0.0; }

    // flx = 1: sweep-1 state of every layer for the back-substitution ([slot][layer][wavelength])
    constexpr int NSLOT = 4 * NB * NB + 5 * NB;
    double *const scr = FLX ? a.scratch + (long)ang * NSLOT * n * a.nwno + w : nullptr;
    auto slot = [&](int sl, int i) -> double & { return scr[((long)sl * n + i) * a.nwno]; };
    double top_zmn[NB], top_zpl[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) top_zmn[r] = top_zpl[r] = 0.0;

    if (!THERMAL && !FLX && a.start_layer > 0) {
        // the layers above start_layer were swept by k_sh4_clear (no cloud there): take its state over
        const double *sp = a.state + w;
        auto ld = [&](int sl) { return sp[(long)sl * a.nwno]; };
        int sl = 0;
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int c2 = 0; c2 < NB; ++c2) R.m[r][c2] = ld(sl++);
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int c2 = 0; c2 < NB; ++c2) pMn.m[r][c2] = ld(sl++);
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int c2 = 0; c2 < NB; ++c2) pPl.m[r][c2] = ld(sl++);
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int c2 = 0; c2 < NB; ++c2) pME.m[r][c2] = ld(sl++);
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int c2 = 0; c2 < NB; ++c2) pPE.m[r][c2] = ld(sl++);
        sl = SHC_SHARED + SHC_STATE * (a.first_angle + ang);
        T = ld(sl++);
        kappa = ld(sl++);
        e_top = ld(sl++);            // unclipped exp(-tau/u0) at the top of start_layer (k_sh4_clear)
        p_og = e_top;                // no cloud above: tau_og = tau
        if (!DRVT && NB == 2) e_top = fmax(e_top, EXP_M35);
#pragma unroll
        for (int r = 0; r < NB; ++r) zeta[r] = ld(sl++);
#pragma unroll
        for (int r = 0; r < NB; ++r) delta[r] = ld(sl++);
#pragma unroll
        for (int r = 0; r < NB; ++r) p_zmn_up[r] = ld(sl++);
#pragma unroll
        for (int r = 0; r < NB; ++r) p_zpl_up[r] = ld(sl++);
    }
    // One layer of the sweep.  Called twice per trip of the loop below: the sweep state of a layer (R, the previous layer's
    // four blocks and particular solutions: 18 doubles) then alternates between two register sets instead of being copied
    // back at the end of every layer (18 v_mov_b64 per layer in the single-body loop; the compiler does not unroll a loop
    // around wave-uniform votes by itself).
    auto layer = [&](const int i) {
        const long o = (long)i * pitch + w;
        const unsigned vo = voff0 + (unsigned)i * pitch8;
        auto LD = [&](const double *base, int below = 0) -> double {      // below = 1: the level under the layer
            if constexpr (FAST) return *(const double *)((const char *)base + (vo + (below ? pitch8 : 0u)));
            else return base[o + (below ? pitch : 0)];
        };
        const double dt = LD(a.dtau), w0 = LD(a.w0), cbo = LD(a.cosb_og);
        // ---- Legendre weights of the phase function ----
        double wsg[NS], wmu[NS];
#pragma unroll
        for (int l = 0; l < NS; ++l) { wsg[l] = 1.0; wmu[l] = 1.0; }
        double psing = 0.0;
        bool nocld = false;          // reflected: no cloud in this layer anywhere in the wave (see below)
        if (!THERMAL) {
            const double fc = LD(a.ftau_cld), fr = LD(a.ftau_ray);
            // f_deltaM as angle k of the reference sees it: its TTHG branch multiplies the array in
            // place by `fac` once per angle (:2823-2824), so the OTHG branch of angle k reads
            // f_deltaM fac^k (the aliasing is live when one form is OTHG and the other TTHG) and
            // the TTHG branch f_deltaM fac^(k+1).
            double fd_prev = LD(a.f_deltaM), fd = fd_prev, f = 0.0, gf = 0.0, gb = 0.0;
            // No cloud in this layer anywhere in the wave and Rayleigh-weighted moments (w_*_rayleigh = 1):
            // every l >= 1 moment is multiplied by ftau_cld = 0, so the weights are (1, 0, ftau_ray/2, 0)
            // whatever the phase-function form and f_deltaM -- the TTHG / OTHG blocks (a pow, two
            // reciprocals, the compounding loop) are skipped.  Same values as the general path gives on
            // such a layer (there the moments come out as +-0).
            nocld = (w_single_rayleigh == 1) && (w_multi_rayleigh == 1) && __all(fc == 0.0);
            const bool tthg = !nocld && (w_single_form == 0 || w_multi_form == 0);
            if (nocld) {
#pragma unroll
                for (int l = 1; l < NS; ++l) { wsg[l] = 0.0; wmu[l] = 0.0; }
                if (NS == 4) { wsg[2] = 0.5 * fr; wmu[2] = 0.5 * fr; }
            }
            if (tthg) {
                gf = a.constant_forward * cbo;
                gb = a.constant_back * cbo;
                f = a.frac_a + a.frac_b * pow_frac(gb, frac_c);
                double cfs = 1.0, cbs = 1.0;
#pragma unroll
                for (int l = 0; l < NS; ++l) { cfs *= a.constant_forward; cbs *= a.constant_back; }
                const double fac = (f * cfs + (1 - f) * cbs);
                for (int p = 1; p < fd_power; ++p) fd_prev *= fac;
                fd = fd_prev * fac;
            }
            if (!nocld && (w_single_form == 1 || w_multi_form == 1)) {   // OTHG :2811-2817
                double cl = 1.0;
                const double ifp = frcp(1 - fd_prev);
#pragma unroll
                for (int l = 1; l < NS; ++l) {
                    cl *= cbo;
                    const double ww = ((2 * l + 1) * cl - (2 * l + 1) * fd_prev) * ifp;
                    if (w_single_form == 1) wsg[l] = ww;
                    if (w_multi_form == 1) wmu[l] = ww;
                }
            }
            if (tthg) {                                                      // TTHG :2819-2831
                double gfl = 1.0, gbl = 1.0;
                const double ifd = frcp(1 - fd);
#pragma unroll
                for (int l = 1; l < NS; ++l) {
                    gfl *= gf;
                    gbl *= gb;
                    const double ww = ((2 * l + 1) * (f * gfl + (1 - f) * gbl) - (2 * l + 1) * fd) * ifd;
                    if (w_single_form == 0) wsg[l] = ww;
                    if (w_multi_form == 0) wmu[l] = ww;
                }
            }
            if (!nocld && w_single_rayleigh == 1) {                        // :2833-2836
#pragma unroll
                for (int l = 1; l < NS; ++l) wsg[l] *= fc;
                if (NS == 4) wsg[2] += 0.5 * fr;
            }
            if (!nocld && w_multi_rayleigh == 1) {                         // :2837-2840
#pragma unroll
                for (int l = 1; l < NS; ++l) wmu[l] *= fc;
                if (NS == 4) wmu[2] += 0.5 * fr;
            }
            if (single_form == 0 && nocld && psingle_rayleigh == 1) {       // ftau_cld (..) + Rayleigh = Rayleigh
                psing = fr * (0.75 * (1 + ct * ct));
            } else if (single_form == 0) {                                 // :2843-2855
                if (psingle_form == 1) psing = hg_term(cbo, ct);
                else if (psingle_form == 0) {
                    const double gf = a.constant_forward * cbo, gb = a.constant_back * cbo;
                    const double f = a.frac_a + a.frac_b * pow_frac(gb, frac_c);
                    psing = f * hg_term(gf, ct) + (1 - f) * hg_term(gb, ct);
                }
                if (psingle_rayleigh == 1) psing = fc * psing + fr * (0.75 * (1 + ct * ct));
            } else {                                                         // legendre form :2954-2957
#pragma unroll
                for (int l = 0; l < NS; ++l) psing += wsg[l] * Pu0[l] * Pu1[l];
            }
        } else {                                                             // thermal :3072-3083
            const double ff = a.use_ff ? ((NS == 4) ? cbo * cbo * cbo * cbo : cbo * cbo) : 0.0;
            double cl = 1.0;
            const double iff = frcp(1 - ff);
#pragma unroll
            for (int l = 0; l < NS; ++l) {
                wmu[l] = (2 * l + 1) * (cl - ff) * iff;
                cl *= cbo;
            }
        }
        double al[NS], bl[NS];
#pragma unroll
        for (int l = 0; l < NS; ++l) {                                       // :2858-2860, :3083
            al[l] = (2 * l + 1) - w0 * wmu[l];
            bl[l] = THERMAL ? 0.0 : (F * (w0 * wsg[l])) * Pu0[l] * (0.25 / PI);
        }
        const double edt_layer = fexp2(dt * g.nl1, K);                   // exp(-dtau/u1)
        // ---- modes ----
        Modes<NB> M;
        constexpr bool BATCH = PZ_SH_BATCH_RCP && NB == 2 && !THERMAL;
        ShRcp rq;
        if constexpr (BATCH) modes_sh4_batched(al, dt, g.iu0, g.iu1, M, rq, K);
        else if constexpr (NB == 2) modes_sh4(al, dt, M, K);
        else modes_sh2(al, dt, M, K);
        // ---- particular solution at the layer top / bottom ----
        double eta[NS];
        double zmn_dn[NB], zpl_dn[NB], zmn_up[NB], zpl_up[NB];
        double B0 = 0.0, b1 = 0.0, ed_layer = 0.0, exp_dt_u0 = 0.0;
        if (!THERMAL) {
            double zpl[NB], zmn[NB];
            double tau_t = 0.0, tau_b = 0.0;
            if constexpr (!DRVT) { tau_t = LD(a.tau); tau_b = LD(a.tau, 1); }
            // DRVT: exp(-dtau/u0), the factor of the running products
            const double fac0 = !DRVT ? 0.0 : (sym ? edt_layer : fexp2(dt * g.nl0, K));
            exp_dt_u0 = fac0;
            double ed, eu;
            if constexpr (NB == 2) {                                         // :3397-3416, :3441-3450
                const double x = iu0, x2 = x * x;
                const double iDel = BATCH ? rq.iDel : frcp(9 * (x2 * x2 - M.beta * x2 + M.gama));
                const double a0 = al[0], a1 = al[1], a2 = al[2], a3 = al[3];
                const double b0 = bl[0], b1_ = bl[1], b2 = bl[2], b3 = bl[3];
                {
                    // eta_l = Delta_l / Delta (:3398-3411), operations written out (no contraction) so that
                    // the cloud-free form below -- the same expressions with the terms that carry
                    // b1 = b3 = 0 (odd moments times ftau_cld = 0) dropped, each dropped term an exact
                    // zero -- gives bit-identical values on such a layer
#pragma clang fp contract(off)
                    const double P = fma(a2, a3, -(9 * x2)), Q = fma(a0, a1, -x2);
                    const double a3b2 = a3 * b2, b0x = b0 * x;
                    double A, B, C, D, E_, F_;
                    if (PZ_SH_OPT_NC && nocld) {
                        A = a1 * b0;
                        B = a3b2 - 2 * (a3 * b0);
                        C = -b0x;
                        D = a3b2;
                        E_ = -(3 * (b2 * x));
                        F_ = -(3 * b0x);
                    } else {
                        const double a0b1 = a0 * b1_, b3x3 = 3 * (b3 * x);
                        A = a1 * b0 - b1_ * x;
                        B = (a3b2 - 2 * (a3 * b0)) - b3x3;
                        C = a0b1 - b0x;
                        D = a3b2 - b3x3;
                        E_ = a2 * b3 - 3 * (b2 * x);
                        F_ = (3 * a0b1 - 2 * (a0 * b3)) - 3 * b0x;
                    }
                    eta[0] = fma(A, P, 2 * (B * x2)) * iDel;
                    eta[1] = fma(C, P, -((2 * a0) * (D * x))) * iDel;
                    eta[2] = fma(D, Q, -((2 * a3) * (C * x))) * iDel;
                    eta[3] = fma(E_, Q, 2 * (F_ * x2)) * iDel;
                    const double h0 = 0.5 * eta[0], e58 = 0.625 * eta[2], m0 = -0.125 * eta[0];
                    zpl[0] = ((h0 + eta[1]) + e58) * (2 * PI);
                    zmn[0] = ((h0 - eta[1]) + e58) * (2 * PI);
                    zpl[1] = ((m0 + e58) + eta[3]) * (2 * PI);
                    zmn[1] = ((m0 + e58) - eta[3]) * (2 * PI);
                }
                if constexpr (DRVT) {
                    ed = fmax(e_top, EXP_M35);
                    e_top = e_top * fac0;                                    // unclipped, carried
                    eu = fmax(e_top, EXP_M35);
                } else {
                ed = (i == 0) ? fexp2_clip(tau_t * g.nl0, K) : e_top;    // = last layer's eu (same element)
                // exp(-tau[i+1]/u0) = exp(-tau[i]/u0) exp(-dtau/u1) in the symmetric geometry when the level
                // depths are the running sums of the layer depths (bit-exact, whole wave) and the 35-clip
                // does not bind at the bottom: one exponential less per layer
                if (PZ_SH_OPT_EXP && sym && __all(tau_b == tau_t + dt) && __all(tau_b * g.nl0 >= -35.0 * LOG2E))
                    eu = ed * edt_layer;
                else
                    eu = fexp2_clip(tau_b * g.nl0, K);
                }
            } else {                                                         // :3240-3265
                const double x = iu0;
                const double iDel = frcp(x * x - al[0] * al[1]);
                eta[0] = (bl[1] * x - al[1] * bl[0]) * iDel;
                eta[1] = (bl[0] * x - al[0] * bl[1]) * iDel;
                zmn[0] = (0.5 * eta[0] - eta[1]) * 2 * PI;
                zpl[0] = (0.5 * eta[0] + eta[1]) * 2 * PI;
                if constexpr (DRVT) {
                    ed = e_top;
                    e_top = e_top * fac0;
                    eu = e_top;
                } else {
                    ed = (i == 0) ? fexp2(tau_t * g.nl0, K) : e_top;
                    eu = fexp2(tau_b * g.nl0, K);
                }
            }
            if constexpr (!DRVT) e_top = eu;
            ed_layer = ed;
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                zmn_dn[r] = zmn[r] * ed; zpl_dn[r] = zpl[r] * ed;
                zmn_up[r] = zmn[r] * eu; zpl_up[r] = zpl[r] * eu;
            }
        } else {                                                             // :3451-3459 / :3266-3270
            B0 = Bn;
            Bn = planck_lambda(a.tlevel[i + 1], wn, K);
            b1 = (Bn - B0) * frcp(dt);
            b1_last = b1;
            const double oma = (1 - w0) * frcp(al[0]), b1a = b1 * frcp(al[1]);
            zmn_dn[0] = oma * (B0 / 2 - b1a) * 2 * PI;
            zpl_dn[0] = oma * (B0 / 2 + b1a) * 2 * PI;
            zmn_up[0] = oma * (B0 / 2 - b1a + b1 * dt / 2) * 2 * PI;
            zpl_up[0] = oma * (B0 / 2 + b1a + b1 * dt / 2) * 2 * PI;
            if constexpr (NB == 2) {
                zmn_dn[1] = zpl_dn[1] = -0.125 * oma * (B0) * 2 * PI;
                zmn_up[1] = zpl_up[1] = -0.125 * oma * (B0 + b1 * dt) * 2 * PI;
            }
#pragma unroll
            for (int l = 0; l < NS; ++l) eta[l] = 0.0;
        }
        const Blk<NB> ME = scale_cols(M.Mn, M.E), PE = scale_cols(M.Pl, M.E);

        // ---- functional weights (source-function integrals) ----
        double gd[NB], gv[NB], c, Tn;
        {
            const double e_u1 = THERMAL ? 0.0 : 0.0;
            (void)e_u1;
            double cm[NS];
            if constexpr (NB == 2) {
                // sum_j w_multi_j P_j(u1) A[j][m] (:3601-3605): A's columns come in +- pairs (rows 1 and 3
                // change sign between the decaying and the growing mode), so the four sums are two even and
                // two odd parts; the odd parts vanish exactly on a cloud-free layer (w_multi_1 = w_multi_3 = 0)
#pragma clang fp contract(off)
                const double wP0 = wmu[0] * Pu1[0], wP2 = wmu[2] * Pu1[2];
                const double e01 = fma(wP2, M.cA[2][0], wP0), e23 = fma(wP2, M.cA[2][2], wP0);
                if (PZ_SH_OPT_NC && !THERMAL && nocld) {
                    cm[0] = cm[1] = e01;
                    cm[2] = cm[3] = e23;
                } else {
                    const double wP1 = wmu[1] * Pu1[1], wP3 = wmu[3] * Pu1[3];
                    const double o01 = fma(wP3, M.cA[3][0], wP1 * M.cA[1][0]);
                    const double o23 = fma(wP3, M.cA[3][2], wP1 * M.cA[1][2]);
                    cm[0] = e01 + o01;
                    cm[1] = e01 - o01;
                    cm[2] = e23 + o23;
                    cm[3] = e23 - o23;
                }
            } else {                                                         // :2916-2917, :3124-3125
                cm[0] = (wmu[0] - wmu[1] * Pu1[1] * M.q);
                cm[1] = (wmu[0] + wmu[1] * Pu1[1] * M.q);
            }
            const double scale = THERMAL ? 2 * PI : 1.0;                     // :3167 vs :2961
            const double tw = T * iu1 * w0 * scale;
            const double edt = edt_layer;                                    // exp(-dtau/u1)
            // exp(-(1/u1 +- lam) dtau) = exp(-dtau/u1) E^{+-1} when no 35-clip binds anywhere in the
            // wave ((1/u1 + lam_max) dtau <= 35 covers all four arguments); otherwise the reference's
            // clipped exponentials are formed directly (:2929-2937).
            // The choice is made per LANE (a wavelength's bits must not depend on which other wavelengths share
            // its wave: blocks of a sharded spectrum cut the waves differently); only the work is wave-uniform --
            // the direct exponentials are formed when some lane of the wave needs them.
            const bool noclip_lane = (iu1 + M.lam[0]) * dt <= 35.0;
            const bool noclip = __all(noclip_lane);
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const double alpha = iu1 + M.lam[r], beta = iu1 - M.lam[r];
                const double rab = BATCH ? rq.rab[r] : frcp(alpha * beta);    // one reciprocal for both
                double ea = edt * M.E[r], eb = edt * (BATCH ? rq.iE[r] : frcp(M.E[r]));
                if (!noclip) {
                    const double ead = fexpk(-clip35(alpha * dt), K), ebd = fexpk(-clip35(beta * dt), K);
                    ea = noclip_lane ? ea : ead;
                    eb = noclip_lane ? eb : ebd;
                }
                const double ha = (1 - ea) * (rab * beta);
                const double hb = (1 - eb) * (rab * alpha);
                gd[r] = tw * cm[2 * r] * ha;
                gv[r] = tw * cm[2 * r + 1] * hb * M.E[r];
            }
            if (!THERMAL) {
                // (1 - exp(-clip35(mus dtau)))/mus (:2901-2905); mus = 2/u1 in the symmetric geometry
                const bool sq_lane = sym && (mus * dt <= 35.0);                 // per lane, as above
                double e_mus = edt * edt;
                if (!__all(sq_lane)) {
                    const double d = fexp2_clip(dt * g.nlm, K);
                    e_mus = sq_lane ? e_mus : d;
                }
                const double exptrm_mus = (1 - e_mus) * imus;
                // exp(-clip35(tau/u0)): the layer-top exponential above (SH4 clips it too; SH2 does not)
                const double expon1 = exptrm_mus * ((NB == 2) ? ed_layer : fmax(ed_layer, EXP_M35));
                double Nsum;
                {                                                            // :2919-2920, 2945-2948
#pragma clang fp contract(off)
                    double sN = (wmu[0] * Pu1[0]) * eta[0];
                    if (NS == 4) sN = fma(wmu[2] * Pu1[2], eta[2], sN);
                    if (!(PZ_SH_OPT_NC && nocld)) {                          // odd moments: zero without cloud
                        sN = fma(wmu[1] * Pu1[1], eta[1], sN);
                        if (NS == 4) sN = fma(wmu[3] * Pu1[3], eta[3], sN);
                    }
                    Nsum = sN * expon1;
                }
                const double dto = LD(a.dtau_og);
                double e_muso = e_mus;
                double faco = 0.0;                                           // DRVT: exp(-dtau_og/u0)
                if constexpr (DRVT) faco = exp_dt_u0;
                if (!__all(dto == dt)) {
                    const double d = fexp2_clip(dto * g.nlm, K);
                    e_muso = (dto == dt) ? e_mus : d;
                    if constexpr (DRVT) {
                        const double d0 = fexp2(dto * g.nl0, K);
                        faco = (dto == dt) ? faco : d0;
                    }
                }
                // exp(-tau_og/u0) at the layer top, unclipped (:2962)
                double e_tauo;
                if constexpr (DRVT) {
                    e_tauo = p_og;
                    p_og = p_og * faco;
                } else {
                // the exponential of tau already at hand when nothing above has been delta-scaled (tau_og == tau in
                // the whole wave)
                const double tauo = LD(a.tau_og);
                e_tauo = (PZ_SH_OPT_EXP && __all(tauo == LD(a.tau)) && __all(tauo * g.nl0 >= -35.0 * LOG2E))
                             ? ed_layer : fexp2(tauo * g.nl0, K);
                }
                const double single = div_const(LD(a.w0_og) * F, FOURPI, R4PI) * psing *
                                      (1 - e_muso) * e_tauo * imus;          // :2959-2965
                c = T * iu1 * (w0 * Nsum + single);
            } else {
                const double ed2 = edt;                                                                  // :3163-3165
                const double edc = (NB == 2) ? fmax(edt, EXP_M35) : edt;   // exp(-clip35(dtau/u1)) :3154 vs :3127
                const double core = (1 - w0) * u1 * frcp(al[0]);
                const double N0 = wmu[0] * (core * (B0 * (1 - edc) + b1 * (u1 - (dt + u1) * edc)));     // :3128, :3155
                const double N1 = wmu[1] * Pu1[1] * (core * (b1 * (1 - edc) * frcp(al[1])));            // :3129, :3156
                c = T * iu1 * (w0 * (N0 + N1) * 2 * PI +
                               2 * PI * (1 - w0) * u1 * (B0 * (1 - ed2) + b1 * (u1 - (dt + u1) * ed2)));
            }
            Tn = T * edt;
        }
        if (i == n - 1) {
            if (!THERMAL) {       // xint[n] = flux_bot/pi = (Pl E d + Mn v + zpl_up)[0]/pi  (:2891, :2967)
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    gd[r] += Tn * (1.0 / PI) * PE.m[0][r];
                    gv[r] += Tn * (1.0 / PI) * M.Mn.m[0][r];
                }
                c += Tn * (1.0 / PI) * zpl_up[0];
            } else {              // :3173-3176
                c += Tn * (a.hard_surface ? Bn * 2 * PI : (Bn + b1 * u1) * 2 * PI);
            }
        }
        // ---- elimination ----
        if (i == 0) {
            double bt[NB];
            double b_top;
            if (!THERMAL) b_top = a.b_top;
            else {
                const double tau_top = dt * a.plevel[0] / (a.plevel[1] - a.plevel[0]);   // :3062-3063
                b_top = PI * (1.0 - fexpk(-tau_top / 0.5, K)) * B_top;
            }
            bt[0] = b_top - zmn_dn[0];                                       // :3479-3480 / :3283
            if constexpr (NB == 2) bt[1] = -b_top / 4 - zmn_dn[1];
            const Blk<NB> Mni = inv(M.Mn);
            R = mm(Mni, PE);
            mv(Mni, bt, delta);
            double tmp[NB];
            mtv(R, gd, tmp);
            kappa = c + dot(gd, delta);
#pragma unroll
            for (int r = 0; r < NB; ++r) zeta[r] = gv[r] - tmp[r];
        } else {
            Blk<NB> A1 = mm(pME, R), A2 = mm(pPE, R);
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
                for (int s = 0; s < NB; ++s) {
                    A1.m[r][s] = pPl.m[r][s] - A1.m[r][s];
                    A2.m[r][s] = pMn.m[r][s] - A2.m[r][s];
                }
            const Blk<NB> A2i = inv(A2);
            const Blk<NB> G = mm(A1, A2i);
            double cP[NB], cM[NB], t1[NB], t2[NB];
            mv(pPE, delta, t1);
            mv(pME, delta, t2);
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                cP[r] = zpl_dn[r] - p_zpl_up[r] - t1[r];
                cM[r] = t2[r] + p_zmn_up[r] - zmn_dn[r];
            }
            Blk<NB> K = mm(G, M.Pl), GM = mm(G, ME);
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
                for (int s = 0; s < NB; ++s) {
                    K.m[r][s] = M.Mn.m[r][s] - K.m[r][s];
                    GM.m[r][s] = PE.m[r][s] - GM.m[r][s];        // -(G Mn E - Pl E)
                }
            K = inv(K);
            const Blk<NB> Rn = mm(K, GM);
            double deltan[NB], rhs[NB];
            mv(G, cP, rhs);
#pragma unroll
            for (int r = 0; r < NB; ++r) rhs[r] += cM[r];
            mv(K, rhs, deltan);
            double tv[NB];
            mv(M.Pl, deltan, tv);
#pragma unroll
            for (int r = 0; r < NB; ++r) tv[r] += cP[r];
            if constexpr (!FLX) {
                // v_{i-1} = Sm v_i + t with Sm = A2i (ME - Pl Rn), t = A2i tv enters the TOA functional only through
                // zeta.t and Sm^T zeta: with z = A2i^T zeta these are z.tv and ME^T z - Rn^T (Pl^T z), and the two
                // NB x NB products that form Sm (and the matrix-vector product for t) are not needed -- 34 instead of
                // 49 fp64 instructions per layer at NB = 2 (the flx = 1 kernels keep Sm and t: their second sweep reads them)
                double z[NB], y[NB], z1[NB], z2[NB];
                mtv(A2i, zeta, z);
                kappa = kappa + dot(z, tv) + dot(gd, deltan) + c;
                mtv(M.Pl, z, y);
#pragma unroll
                for (int r = 0; r < NB; ++r) y[r] += gd[r];
                mtv(ME, z, z1);
                mtv(Rn, y, z2);
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    zeta[r] = z1[r] + gv[r] - z2[r];
                    delta[r] = deltan[r];
                }
                R = Rn;
            } else {
            Blk<NB> Sm = mm(M.Pl, Rn);
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
                for (int s = 0; s < NB; ++s) Sm.m[r][s] = ME.m[r][s] - Sm.m[r][s];
            Sm = mm(A2i, Sm);
            double t[NB];
            mv(A2i, tv, t);
            if constexpr (FLX) {                       // v_{i-1} = Sm v_i + t
                int sl = 3 * NB * NB + 4 * NB;
#pragma unroll
                for (int r = 0; r < NB; ++r)
#pragma unroll
                    for (int c2 = 0; c2 < NB; ++c2) slot(sl++, i) = Sm.m[r][c2];
#pragma unroll
                for (int r = 0; r < NB; ++r) slot(sl++, i) = t[r];
            }
            kappa = kappa + dot(zeta, t) + dot(gd, deltan) + c;
            double z1[NB], z2[NB];
            mtv(Sm, zeta, z1);
            mtv(Rn, gd, z2);
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                zeta[r] = z1[r] + gv[r] - z2[r];
                delta[r] = deltan[r];
            }
            R = Rn;
            }
        }
        if constexpr (FLX) {
            int sl = 0;
#pragma unroll
            for (int r = 0; r < NB; ++r) slot(sl++, i) = delta[r];
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
                for (int c = 0; c < NB; ++c) slot(sl++, i) = R.m[r][c];
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
                for (int c = 0; c < NB; ++c) slot(sl++, i) = M.Mn.m[r][c];
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
                for (int c = 0; c < NB; ++c) slot(sl++, i) = M.Pl.m[r][c];
#pragma unroll
            for (int r = 0; r < NB; ++r) slot(sl++, i) = M.E[r];
#pragma unroll
            for (int r = 0; r < NB; ++r) slot(sl++, i) = zmn_up[r];
#pragma unroll
            for (int r = 0; r < NB; ++r) slot(sl++, i) = zpl_up[r];
            if (i == 0) {
#pragma unroll
                for (int r = 0; r < NB; ++r) { top_zmn[r] = zmn_dn[r]; top_zpl[r] = zpl_dn[r]; }
            }
        }
        pMn = M.Mn; pPl = M.Pl; pME = ME; pPE = PE;
#pragma unroll
        for (int r = 0; r < NB; ++r) { p_zmn_up[r] = zmn_up[r]; p_zpl_up[r] = zpl_up[r]; }
        T = Tn;
    };
    if constexpr (FAST && !THERMAL && !FLX) {      // (the generic and flx = 1 bodies are larger: two copies would spill)
        int i = a.start_layer;
        for (; i + 1 < n; i += 2) {
            layer(i);
            layer(i + 1);
        }
        if (i < n) layer(i);
    } else {
        for (int i = (!THERMAL && !FLX) ? a.start_layer : 0; i < n; ++i) layer(i);
    }
    // ---- surface rows (:3484-3494 / :3287-3289) ----
    double bs[NB];
    if (!THERMAL) {
        double e_bot;                                                        // exp(-tau[n]/u0), unclipped
        if constexpr (DRVT) e_bot = e_top;
        else e_bot = fexpk(-a.tau[(long)n * pitch + w] / u0, K);
        const double bsf = (0. + rs * u0 * F * e_bot);                                          // :2863-2864
        bs[0] = bsf;
        if constexpr (NB == 2) bs[1] = -bsf / 4;
    } else {
        bs[0] = a.hard_surface ? PI * Bn : PI * (Bn + b1_last * 0.5);                       // :3065-3068
        if constexpr (NB == 2) bs[1] = (-PI * Bn / 4);                                       // :3070
    }
    Blk<NB> L, W;
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            W.m[r][s] = pPE.m[r][s] - rs * pME.m[r][s];
            L.m[r][s] = pMn.m[r][s] - rs * pPl.m[r][s];
        }
    const Blk<NB> WR = mm(W, R);
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
        for (int s = 0; s < NB; ++s) L.m[r][s] -= WR.m[r][s];
    double wd[NB], rhs[NB], v[NB];
    mv(W, delta, wd);
#pragma unroll
    for (int r = 0; r < NB; ++r) rhs[r] = bs[r] - p_zpl_up[r] + rs * p_zmn_up[r] - wd[r];
    mv(inv(L), rhs, v);
    xint[w] = kappa + dot(zeta, v);
    if constexpr (FLX) {
        // ---- sweep 2: back-substitute v_i, d_i = delta_i - R_i v_i and write the moment fluxes at the
        // bottom of every layer (and the top of layer 0) in the row order of the reference's F.X + G:
        // (Fmn[0..NB), Fpl[0..NB)) per level (fluxes.py:3311-3331, 3552-3599)
        double *fl = a.flux + (long)ang * NS * (n + 1) * a.nwno + w;
        double vv[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) vv[r] = v[r];
        for (int i = n - 1; i >= 0; --i) {
            int sl = 0;
            double dl[NB], E[NB], zmu[NB], zpu[NB], d[NB], Rv[NB];
            Blk<NB> Ri, Mn, Pl;
#pragma unroll
            for (int r = 0; r < NB; ++r) dl[r] = slot(sl++, i);
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
                for (int c2 = 0; c2 < NB; ++c2) Ri.m[r][c2] = slot(sl++, i);
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
                for (int c2 = 0; c2 < NB; ++c2) Mn.m[r][c2] = slot(sl++, i);
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
                for (int c2 = 0; c2 < NB; ++c2) Pl.m[r][c2] = slot(sl++, i);
#pragma unroll
            for (int r = 0; r < NB; ++r) E[r] = slot(sl++, i);
#pragma unroll
            for (int r = 0; r < NB; ++r) zmu[r] = slot(sl++, i);
#pragma unroll
            for (int r = 0; r < NB; ++r) zpu[r] = slot(sl++, i);
            mv(Ri, vv, Rv);
#pragma unroll
            for (int r = 0; r < NB; ++r) d[r] = dl[r] - Rv[r];
            const Blk<NB> MEi = scale_cols(Mn, E), PEi = scale_cols(Pl, E);
            double x1[NB], x2[NB];
            mv(MEi, d, x1);
            mv(Pl, vv, x2);
#pragma unroll
            for (int r = 0; r < NB; ++r) fl[(long)(NS * (i + 1) + r) * a.nwno] = x1[r] + x2[r] + zmu[r];
            mv(PEi, d, x1);
            mv(Mn, vv, x2);
#pragma unroll
            for (int r = 0; r < NB; ++r) fl[(long)(NS * (i + 1) + NB + r) * a.nwno] = x1[r] + x2[r] + zpu[r];
            if (i == 0) {
                mv(Mn, d, x1);
                mv(PEi, vv, x2);
#pragma unroll
                for (int r = 0; r < NB; ++r) fl[(long)r * a.nwno] = x1[r] + x2[r] + top_zmn[r];
                mv(Pl, d, x1);
                mv(MEi, vv, x2);
#pragma unroll
                for (int r = 0; r < NB; ++r) fl[(long)(NB + r) * a.nwno] = x1[r] + x2[r] + top_zpl[r];
            } else {
                Blk<NB> Smi;
                double ti[NB], nv[NB];
#pragma unroll
                for (int r = 0; r < NB; ++r)
#pragma unroll
                    for (int c2 = 0; c2 < NB; ++c2) Smi.m[r][c2] = slot(sl++, i);
#pragma unroll
                for (int r = 0; r < NB; ++r) ti[r] = slot(sl++, i);
                mv(Smi, vv, nv);
#pragma unroll
                for (int r = 0; r < NB; ++r) vv[r] = nv[r] + ti[r];
            }
        }
    }
}

template <int NB, bool THERMAL, bool FLX, bool FAST = false, bool DRVT = false>
__global__ __launch_bounds__(256, PZ_SH_MINWAVES) void k_sh(const SHArgs a)
{
    sh_body<NB, THERMAL, FLX, FAST, DRVT>(a, a.ang);
}

// Reflected light without layer fluxes: ONE kernel for a single spectrum and for `nspec` spectra of one shape and option
// set in one grid (picaso_get_reflected_SH_batch_dev: blockIdx.y = spectrum, whose planes, output and angle table come
// from the device table a.batch, read through the constant address space like kernel arguments).  A single launch is the
// same machine code with a.batch = NULL -- its angle table is the kernel-argument segment itself -- so "spectrum s of a
// batch is bit-identical to its own launch" does not rest on the compiler contracting two instantiations of the body
// alike (round 5: it did not, once the layer body was inlined twice).
template <int NB, bool FAST, bool DRVT = false>
__global__ __launch_bounds__(256, PZ_SH_MINWAVES) void k_sh_refl(const SHArgs a)
{
    typedef const __attribute__((address_space(4))) SHBatchItem *ItemPtr;
    typedef const __attribute__((address_space(4))) SHArgs::Angle *AngPtr;
    typedef const __attribute__((address_space(4))) char *ArgBytes;
    SHArgs b = a;
    AngPtr angp = (AngPtr)((ArgBytes)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(SHArgs, ang));
    if (a.batch) {
        const ItemPtr it = (ItemPtr)(unsigned long)a.batch + blockIdx.y;
        b.dtau = it->dtau; b.tau = it->tau; b.w0 = it->w0; b.ftau_cld = it->ftau_cld; b.ftau_ray = it->ftau_ray;
        b.f_deltaM = it->f_deltaM; b.dtau_og = it->dtau_og; b.tau_og = it->tau_og; b.w0_og = it->w0_og;
        b.cosb_og = it->cosb_og; b.surf_reflect = it->surf_reflect; b.F0PI = it->F0PI; b.xint = it->xint;
        b.cos_theta = it->cos_theta;
        angp = it->ang;
    }
    sh_body<NB, false, false, FAST, DRVT>(b, angp);
}


// ---------------------------------------------------------------------------------------------------------------
// Thermal emission: the angle-independent block algebra shared between the disk angles of a lane.
//
// In get_thermal_SH (fluxes.py:2979-3182) the angle enters only through the source-function integration along ubar1
// (:3105-3182): the Legendre weights, the stream matrices, the particular solution and therefore the whole sweep
// relation d_i = delta_i - R_i v_i are the same for every (g, t).  k_sh<NB, true> nevertheless redid them per angle
// (one wave per angle: 5 x 560 fp64 instructions per wavelength-layer at five angles).  Here one lane carries NA
// angles: per layer the modes, the Planck terms and the elimination once (~400 instructions), and per angle only
// exp(-dtau/ubar1), the integration weights (gd, gv, c) and the update of its TOA functional (kappa, zeta, T):
// ~110 instructions each.  Small launches keep fewer angles per lane (down to one) so that every SIMD has a wave;
// the arithmetic below is written out (contraction off, explicit fma) so that an angle's result does not depend on
// how many angles share its lane.
// ---------------------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)

struct SHTArgs {
    int nlayer, nwno;
    long pitch;
    const double *dtau, *w0, *cosb_og, *surf_reflect, *wno, *tlevel, *plevel;
    int hard_surface, use_ff;
    int na;                                         // angles of this launch: blockIdx.y = chunk of NA of them
    struct Angle { double u1, iu1, nl1, p1, p2, p3; } ang[SH_MAX_ANG];
    double *xint;                                   // first angle of this launch, (na, nwno)
};

template <int NB>
__device__ __forceinline__ Blk<NB> mmx(const Blk<NB> &A, const Blk<NB> &B)
{
    Blk<NB> C;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            double s = A.m[i][0] * B.m[0][j];
#pragma unroll
            for (int k = 1; k < NB; ++k) s = fma(A.m[i][k], B.m[k][j], s);
            C.m[i][j] = s;
        }
    return C;
}
template <int NB>
__device__ __forceinline__ void mvx(const Blk<NB> &A, const double (&x)[NB], double (&y)[NB])
{
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        double s = A.m[i][0] * x[0];
#pragma unroll
        for (int k = 1; k < NB; ++k) s = fma(A.m[i][k], x[k], s);
        y[i] = s;
    }
}
template <int NB>
__device__ __forceinline__ void mtvx(const Blk<NB> &A, const double (&x)[NB], double (&y)[NB])   // A^T x
{
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        double s = A.m[0][i] * x[0];
#pragma unroll
        for (int k = 1; k < NB; ++k) s = fma(A.m[k][i], x[k], s);
        y[i] = s;
    }
}
template <int NB>
__device__ __forceinline__ double dotx(const double (&a)[NB], const double (&b)[NB])
{
    double s = a[0] * b[0];
#pragma unroll
    for (int i = 1; i < NB; ++i) s = fma(a[i], b[i], s);
    return s;
}
template <int NB>
__device__ __forceinline__ Blk<NB> invx(const Blk<NB> &A)
{
    Blk<NB> R;
    if (NB == 1) {
        R.m[0][0] = frcp(A.m[0][0]);
    } else {
        const double idet = frcp(fma(A.m[0][0], A.m[NB - 1][NB - 1], -(A.m[0][NB - 1] * A.m[NB - 1][0])));
        R.m[0][0] = A.m[NB - 1][NB - 1] * idet;
        R.m[NB - 1][NB - 1] = A.m[0][0] * idet;
        R.m[0][NB - 1] = -(A.m[0][NB - 1] * idet);
        R.m[NB - 1][0] = -(A.m[NB - 1][0] * idet);
    }
    return R;
}
template <int NB>
__device__ __forceinline__ Blk<NB> subx(const Blk<NB> &A, const Blk<NB> &B)
{
    Blk<NB> C;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) C.m[i][j] = A.m[i][j] - B.m[i][j];
    return C;
}

// modes_sh4 / modes_sh2 with the arithmetic written out (fluxes.py:3388-3434, 3245-3251)
struct ModesT4 { double lam[2], E[2], iE[2], R[2], Q[2], S[2], beta, gama; Blk<2> Mn, Pl; };
__device__ __forceinline__ void modes_sh4x(const double (&a)[4], double dt, ModesT4 &M, const Exp2Coef &K)
{
    const double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
    const double a01 = a0 * a1;
    const double beta = (a01 + div_const(4 * (a0 * a3), 9.0, R9)) + div_const(a2 * a3, 9.0, R9);
    const double gama = div_const((a01 * a2) * a3, 9.0, R9);
    M.beta = beta;
    M.gama = gama;
    const double disc = fsqrt(fma(beta, beta, -(4 * gama)));
    const double x1 = 0.5 * (beta + disc), x2 = 0.5 * (beta - disc);
    const double il1 = frsq(x1), il2 = frsq(x2);
    const double l1 = x1 * il1, l2 = x2 * il2;
    M.lam[0] = l1;
    M.lam[1] = l2;
    const double s3 = -1.5 * frcp(a3);
    const double R1 = -(a0 * il1), R2 = -(a0 * il2);
    const double Q1 = 0.5 * fma(a01 * il1, il1, -1.0), Q2 = 0.5 * fma(a01 * il2, il2, -1.0);
    const double S1 = s3 * fma(a01, il1, -l1), S2 = s3 * fma(a01, il2, -l2);
    M.R[0] = R1; M.R[1] = R2; M.Q[0] = Q1; M.Q[1] = Q2; M.S[0] = S1; M.S[1] = S2;
    const double tp = 2 * PI, q1 = 0.625 * Q1, q2 = 0.625 * Q2;
    M.Pl.m[0][0] = ((0.5 + R1) + q1) * tp;
    M.Pl.m[0][1] = ((0.5 + R2) + q2) * tp;
    M.Pl.m[1][0] = ((-0.125 + q1) + S1) * tp;
    M.Pl.m[1][1] = ((-0.125 + q2) + S2) * tp;
    M.Mn.m[0][0] = ((0.5 - R1) + q1) * tp;
    M.Mn.m[0][1] = ((0.5 - R2) + q2) * tp;
    M.Mn.m[1][0] = ((-0.125 + q1) - S1) * tp;
    M.Mn.m[1][1] = ((-0.125 + q2) - S2) * tp;
    M.E[0] = fexpk(-clip35(l1 * dt), K);
    M.E[1] = fexpk(-clip35(l2 * dt), K);
    M.iE[0] = frcp(M.E[0]);
    M.iE[1] = frcp(M.E[1]);
}
struct ModesT2 { double lam[1], E[1], iE[1], q; Blk<1> Mn, Pl; };
__device__ __forceinline__ void modes_sh2x(const double (&a)[2], double dt, ModesT2 &M, const Exp2Coef &K)
{
    const double lam = fsqrt(a[0] * a[1]);
    M.lam[0] = lam;
    M.q = lam * frcp(a[1]);
    M.Mn.m[0][0] = (0.5 + M.q) * (2 * PI);
    M.Pl.m[0][0] = (0.5 - M.q) * (2 * PI);
    M.E[0] = fexpk(-clip35(lam * dt), K);
    M.iE[0] = frcp(M.E[0]);
}

template <int NB, int NA>
__global__ __launch_bounds__(256, 2) void k_sh_thermal(const SHTArgs a)
{
    constexpr int NS = 2 * NB;
    using ModesT = typename std::conditional<NB == 2, ModesT4, ModesT2>::type;
    const long w = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.nwno) return;
    const int n = a.nlayer;
    const long pitch = a.pitch;
    const int k0 = blockIdx.y * NA;                 // this lane's angles: k0 .. k0 + na - 1 (the last chunk may be short)
    const int na = min(NA, a.na - k0);
    const SHTArgs::Angle *const ang = a.ang + k0;
    Exp2Coef K;
    K.load();
    const double rs = a.surf_reflect[w], wn = a.wno[w];
    // Planck function at the levels (fluxes.py:3058-3060): planck_lambda with its level-independent factors hoisted
    const double hP = 6.62607004e-27, cP_ = 2.99792458e+10, kP = 1.38064852e-16;
    const double wcm = 1.0 / wn, w2 = wcm * wcm;
    const double pf = (2.0 * hP * (cP_ * cP_)) / (w2 * w2 * wcm), wk = wcm * kP, hc = hP * cP_;
    auto planck = [&](double t) { return pf * planck_rcp(fexpk(fdiv(hc, t * wk), K)); };
    double Bn = planck(a.tlevel[0]);
    const double B_top = Bn;
    double b1 = 0.0;
    // per angle: transmission to the top, TOA functional J = kappa + zeta . v
    double T[NA], kappa[NA], zeta[NA][NB];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        T[k] = 1.0;
        kappa[k] = 0.0;
#pragma unroll
        for (int r = 0; r < NB; ++r) zeta[k][r] = 0.0;
    }
    double delta[NB];
    Blk<NB> R, pMn, pPl, pME, pPE;
    double p_zmn_up[NB], p_zpl_up[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) delta[r] = p_zmn_up[r] = p_zpl_up[r] = 0.0;

    for (int i = 0; i < n; ++i) {
        const long o = (long)i * pitch + w;
        const double dt = a.dtau[o], w0 = a.w0[o], cbo = a.cosb_og[o];
        // ---- Legendre weights and stream coefficients (:3072-3083) ----
        double wmu[NS], al[NS];
        {
            const double c2 = cbo * cbo;
            const double ff = a.use_ff ? ((NS == 4) ? c2 * c2 : c2) : 0.0;
            const double iff = frcp(1 - ff);
            double cl = 1.0;
#pragma unroll
            for (int l = 0; l < NS; ++l) {
                wmu[l] = ((2 * l + 1) * (cl - ff)) * iff;
                al[l] = fma(-w0, wmu[l], (double)(2 * l + 1));
                cl *= cbo;
            }
        }
        ModesT M;
        if constexpr (NB == 2) modes_sh4x(al, dt, M, K);
        else modes_sh2x(al, dt, M, K);
        // ---- particular solution (:3451-3459 / :3266-3270) ----
        const double B0 = Bn;
        Bn = planck(a.tlevel[i + 1]);
        b1 = (Bn - B0) * frcp(dt);
        const double omw = 1 - w0;
        const double ia0 = frcp(al[0]), ia1 = frcp(al[1]);
        const double oma = omw * ia0, b1a = b1 * ia1, hB = 0.5 * B0, hbd = 0.5 * (b1 * dt);
        double zmn_dn[NB], zpl_dn[NB], zmn_up[NB], zpl_up[NB];
        zmn_dn[0] = (oma * (hB - b1a)) * (2 * PI);
        zpl_dn[0] = (oma * (hB + b1a)) * (2 * PI);
        zmn_up[0] = (oma * ((hB - b1a) + hbd)) * (2 * PI);
        zpl_up[0] = (oma * ((hB + b1a) + hbd)) * (2 * PI);
        if constexpr (NB == 2) {
            zmn_dn[1] = zpl_dn[1] = ((-0.125 * oma) * B0) * (2 * PI);
            zmn_up[1] = zpl_up[1] = ((-0.125 * oma) * fma(b1, dt, B0)) * (2 * PI);
        }
        const Blk<NB> ME = scale_cols(M.Mn, M.E), PE = scale_cols(M.Pl, M.E);

        // ---- elimination (angle independent) ----
        Blk<NB> Rn, Sm;
        double deltan[NB], t[NB];
        if (i == 0) {
            const double tau_top = dt * a.plevel[0] / (a.plevel[1] - a.plevel[0]);   // :3062-3063
            const double b_top = PI * (1.0 - fexpk(-tau_top / 0.5, K)) * B_top;
            double bt[NB];
            bt[0] = b_top - zmn_dn[0];                                               // :3479-3480 / :3283
            if constexpr (NB == 2) bt[1] = -b_top / 4 - zmn_dn[1];
            const Blk<NB> Mni = invx(M.Mn);
            Rn = mmx(Mni, PE);
            mvx(Mni, bt, deltan);
#pragma unroll
            for (int r = 0; r < NB; ++r) t[r] = 0.0;
            Sm = Rn;                                       // not used for i == 0
        } else {
            const Blk<NB> A1 = subx(pPl, mmx(pME, R)), A2 = subx(pMn, mmx(pPE, R));
            const Blk<NB> A2i = invx(A2);
            const Blk<NB> G = mmx(A1, A2i);
            double cP[NB], cM[NB], t1[NB], t2[NB];
            mvx(pPE, delta, t1);
            mvx(pME, delta, t2);
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                cP[r] = (zpl_dn[r] - p_zpl_up[r]) - t1[r];
                cM[r] = (t2[r] + p_zmn_up[r]) - zmn_dn[r];
            }
            const Blk<NB> Ki = invx(subx(M.Mn, mmx(G, M.Pl)));
            const Blk<NB> GM = subx(PE, mmx(G, ME));
            Rn = mmx(Ki, GM);
            double rhs[NB];
            mvx(G, cP, rhs);
#pragma unroll
            for (int r = 0; r < NB; ++r) rhs[r] += cM[r];
            mvx(Ki, rhs, deltan);
            Sm = mmx(A2i, subx(ME, mmx(M.Pl, Rn)));
            double tv[NB];
            mvx(M.Pl, deltan, tv);
#pragma unroll
            for (int r = 0; r < NB; ++r) tv[r] += cP[r];
            mvx(A2i, tv, t);
        }

        // ---- per angle: source-function integrals of this layer and the update of the TOA functional ----
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            if (k >= na) break;
            const double u1 = ang[k].u1, iu1 = ang[k].iu1;
            const double edt = fexp2(dt * ang[k].nl1, K);                         // exp(-dtau/u1)
            double cm[NS];
            if constexpr (NB == 2) {                                                 // :3601-3605
                const double wP2 = wmu[2] * ang[k].p2, wP1 = wmu[1] * ang[k].p1, wP3 = wmu[3] * ang[k].p3;
                const double e01 = fma(wP2, M.Q[0], wmu[0]), e23 = fma(wP2, M.Q[1], wmu[0]);
                const double o01 = fma(wP3, M.S[0], wP1 * M.R[0]), o23 = fma(wP3, M.S[1], wP1 * M.R[1]);
                cm[0] = e01 + o01;
                cm[1] = e01 - o01;
                cm[2] = e23 + o23;
                cm[3] = e23 - o23;
            } else {                                                                 // :3124-3125
                const double wq = (wmu[1] * ang[k].p1) * M.q;
                cm[0] = wmu[0] - wq;
                cm[1] = wmu[0] + wq;
            }
            const double Tk = T[k];
            const double tw = ((Tk * iu1) * w0) * (2 * PI);                          // :3167
            const bool noclip_lane = (iu1 + M.lam[0]) * dt <= 35.0;      // per lane; the work is wave-uniform (see k_sh)
            const bool noclip = __all(noclip_lane);
            double gd[NB], gv[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const double alpha = iu1 + M.lam[r], beta = iu1 - M.lam[r];
                const double rab = frcp(alpha * beta);
                double ea = edt * M.E[r], eb = edt * M.iE[r];
                if (!noclip) {
                    const double ead = fexpk(-clip35(alpha * dt), K), ebd = fexpk(-clip35(beta * dt), K);
                    ea = noclip_lane ? ea : ead;
                    eb = noclip_lane ? eb : ebd;
                }
                const double ha = (1 - ea) * (rab * beta), hb = (1 - eb) * (rab * alpha);
                gd[r] = (tw * cm[2 * r]) * ha;
                gv[r] = ((tw * cm[2 * r + 1]) * hb) * M.E[r];
            }
            const double edc = (NB == 2) ? fmax(edt, EXP_M35) : edt;               // exp(-clip35(dtau/u1)) :3154 vs :3127
            const double core = oma * u1;
            const double dtu = dt + u1;
            const double N0 = wmu[0] * (core * fma(B0, 1 - edc, b1 * fma(-dtu, edc, u1)));     // :3128, :3155
            const double N1 = (wmu[1] * ang[k].p1) * (core * ((b1 * (1 - edc)) * ia1));      // :3129, :3156
            const double direct = ((2 * PI) * omw) * (u1 * fma(B0, 1 - edt, b1 * fma(-dtu, edt, u1)));   // :3163-3165
            double c = (Tk * iu1) * fma(w0 * (N0 + N1), 2 * PI, direct);
            const double Tn = Tk * edt;
            if (i == n - 1)                                                          // :3173-3176
                c = fma(Tn, a.hard_surface ? Bn * (2 * PI) : fma(b1, u1, Bn) * (2 * PI), c);
            double z2[NB];
            mtvx(Rn, gd, z2);
            if (i == 0) {
                kappa[k] = c + dotx(gd, deltan);
#pragma unroll
                for (int r = 0; r < NB; ++r) zeta[k][r] = gv[r] - z2[r];
            } else {
                kappa[k] = ((kappa[k] + dotx(zeta[k], t)) + dotx(gd, deltan)) + c;
                double z1[NB];
                mtvx(Sm, zeta[k], z1);
#pragma unroll
                for (int r = 0; r < NB; ++r) zeta[k][r] = (z1[r] + gv[r]) - z2[r];
            }
            T[k] = Tn;
        }
        R = Rn;
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            delta[r] = deltan[r];
            p_zmn_up[r] = zmn_up[r];
            p_zpl_up[r] = zpl_up[r];
        }
        pMn = M.Mn; pPl = M.Pl; pME = ME; pPE = PE;
    }
    // ---- surface rows (:3484-3494 / :3287-3289) ----
    double bs[NB];
    bs[0] = a.hard_surface ? PI * Bn : PI * fma(b1, 0.5, Bn);                        // :3065-3068
    if constexpr (NB == 2) bs[1] = -(PI * Bn) / 4;                                   // :3070
    Blk<NB> W, L;
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
        for (int s_ = 0; s_ < NB; ++s_) {
            W.m[r][s_] = fma(-rs, pME.m[r][s_], pPE.m[r][s_]);
            L.m[r][s_] = fma(-rs, pPl.m[r][s_], pMn.m[r][s_]);
        }
    L = subx(L, mmx(W, R));
    double wd[NB], rhs[NB], v[NB];
    mvx(W, delta, wd);
#pragma unroll
    for (int r = 0; r < NB; ++r) rhs[r] = ((bs[r] - p_zpl_up[r]) + rs * p_zmn_up[r]) - wd[r];
    mvx(invx(L), rhs, v);
#pragma unroll
    for (int k = 0; k < NA; ++k)
        if (k < na) a.xint[(long)(k0 + k) * a.nwno + w] = kappa[k] + dotx(zeta[k], v);
}
#pragma clang fp contract(fast)

template <int NB>
static void launch_sht_na(picaso_ctx *ctx, const SHTArgs &a, int na, int nchunk)
{
    const dim3 grid((unsigned)((a.nwno + 255) / 256), (unsigned)nchunk), block(256);
    switch (na) {
    case 1: hipLaunchKernelGGL((k_sh_thermal<NB, 1>), grid, block, 0, ctx->stream, a); break;
    case 2: hipLaunchKernelGGL((k_sh_thermal<NB, 2>), grid, block, 0, ctx->stream, a); break;
    case 3: hipLaunchKernelGGL((k_sh_thermal<NB, 3>), grid, block, 0, ctx->stream, a); break;
    case 4: hipLaunchKernelGGL((k_sh_thermal<NB, 4>), grid, block, 0, ctx->stream, a); break;
    default: hipLaunchKernelGGL((k_sh_thermal<NB, 5>), grid, block, 0, ctx->stream, a); break;
    }
}

// Angles per lane for a launch of `ncol` columns and `nang` angles: the split into ceil(nang / m) nearly equal
// chunks with the smallest modelled time.  Per wave and layer: S shared + m A per-angle instructions (400 / 110,
// counted from the source for SH4); one wave alone on its SIMD issues an fp64 instruction every ~5.5 cycles, two
// or more share the 4 cycles of the pipe, and a launch whose waves are not a multiple of the 2 x SIMD slots pays
// for the partly filled last round.
static int sh_angles_per_lane(const picaso_ctx *ctx, long ncol, int nang, double S, double A, const char *env,
                              int max_per_lane = 5)
{
    if (const char *e = getenv(env)) {
        const int m = atoi(e);
        if (m >= 1 && m <= max_per_lane) return m;
    }
    const long simds = 4L * ctx->ncu, groups = (ncol + 63) / 64;
    int best = 1;
    double best_t = 1e300;
    for (int m = 1; m <= max_per_lane; ++m) {
        const int nchunk = (nang + m - 1) / m;
        const long waves = groups * nchunk;
        const double per_wave = S + A * ((double)nang / nchunk);      // mean chunk
        const double longest = S + A * ((nang + nchunk - 1) / nchunk);
        const long full = waves / (2 * simds), rem = waves - full * 2 * simds;
        const double t = full * 8.0 * per_wave + (rem == 0 ? 0.0 : rem <= simds ? 5.5 * (full ? per_wave : longest)
                                                                                    : 8.0 * per_wave);
        if (t < best_t) { best_t = t; best = m; }
    }
    return best;
}

static int launch_sh_thermal(picaso_ctx *ctx, SHTArgs &a, int stream, int nang, const double *ubar1, double *xint)
{
    const int m = sh_angles_per_lane(ctx, a.nwno, nang, 400.0, 110.0, "PICASO_AMD_SHT_ANGLES");
    const int nchunk = (nang + m - 1) / m;
    const int per_lane = (nang + nchunk - 1) / nchunk;                           // nearly equal chunks, last one short
    const int per_launch = (SH_MAX_ANG / per_lane) * per_lane;                   // whole chunks per launch
    for (int done = 0; done < nang; done += per_launch) {
        const int na = (nang - done < per_launch) ? nang - done : per_launch;
        for (int k = 0; k < na; ++k) {
#pragma clang fp contract(off)
            SHTArgs::Angle &g = a.ang[k];
            const double mu = ubar1[done + k];
            g.u1 = mu;
            g.iu1 = 1.0 / mu;
            g.nl1 = -LOG2E * g.iu1;
            g.p1 = mu;                                                             // legP (fluxes.py:3639-3646)
            g.p2 = (3 * mu * mu - 1) / 2;
            g.p3 = (5 * mu * mu * mu - 3 * mu) / 2;
        }
        a.na = na;
        a.xint = xint + (size_t)done * a.nwno;
        const int chunks = (na + per_lane - 1) / per_lane;
        if (stream == 4) launch_sht_na<2>(ctx, a, per_lane, chunks);
        else launch_sht_na<1>(ctx, a, per_lane, chunks);
        PZ_HIP(ctx, hipGetLastError());
    }
    return 0;
}

static SHArgs::Angle make_sh_angle(double u0, double u1, bool thermal);

// ---------------------------------------------------------------------------------------------------------------
// Reflected light, SH4, cloud-free columns, the reference's default options: the angle-independent half of a layer
// shared between the disk angles of a lane (review item: "measure the angle sharing instead of modelling it").
//
// Without cloud every l >= 1 moment of the phase function is multiplied by ftau_cld = 0 (w_*_rayleigh = 1), so the
// Legendre weights are (1, 0, ftau_ray / 2, 0) with ftau_ray = 1, cosb = 0 makes the delta-scaling the identity
// (f_deltaM = 0: dtau_og = dtau, w0_og = w0, tau_og = tau -- and the reference's in-place compounding of f_deltaM,
// which gives every angle of a cloudy column its own matrices, multiplies a zero), and the level depths are the
// running sums of dtau.  The stream coefficients a_l = (1 - w0, 3, 5 - w0 / 2, 7), the two modes, the blocks Mn / Pl
// and the matrix half of the sweep (R, S, the two inverses) are then the same for every angle; what is left per
// angle is the particular solution along 1/u0, the right-hand sides of the elimination, the source-function weights
// along u1 and the update of its TOA functional: ~190 fp64 instructions per layer shared, ~275 per angle (from the
// timings below) -- k_sh<2,false,false,true> spends ~590 per angle.  The launch reads dtau and w0 only (and F0PI,
// surf_reflect): picaso_get_reflected_SH_dev with the nine other plane arguments NULL.
//
// Same formulas in the same order as sh_body (fluxes.py:2675-2976, 3336-3607) with the exact zeros dropped, written
// out like k_sh_thermal (contraction off, explicit fma): a column's result depends neither on the launch shape nor on
// how many angles share its lane.  It is NOT bit-identical to k_sh on the full plane set of the same column -- k_sh
// takes six reciprocals of a layer, two of them along the angle, from one Newton core (modes_sh4_batched), so even
// its angle-independent blocks carry the angle in their last bits; the two agree to <= 1e-9 like each does with the
// oracle (observed 5e-10 over 1e5 columns; tests/test_sh_clear_gpu.py).
// ---------------------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)

struct SHCArgs {
    int nlayer, nwno;
    long pitch;
    const double *dtau, *w0, *surf_reflect, *F0PI;
    int na;                                         // angles of this launch: blockIdx.y = chunk of NA of them
    struct Angle {
        double u0, iu0, iu1, mus, imus, nl0, nl1, nlm;   // as SHArgs::Angle
        double x2, x4;                              // (1/u0)^2, (1/u0)^4
        double p2u0, hp2u1;                         // P_2(-u0); P_2(u1) / 2
        int sym;                                    // u0 == u1
    } ang[SH_MAX_ANG];
    double psing;                                   // Rayleigh single scattering 0.75 (1 + cos_theta^2), :2846
    double b_top;
    double *xint;                                   // first angle of this launch, (na, nwno)
    // a cloud-free TOP only (picaso_get_reflected_SH_top_dev): sweep the first `nstop` layers and leave the state for
    // k_sh (SHArgs::state) instead of closing the column with the surface rows
    int nstop, first_angle;
    double *state;
};

// Angles per lane: two.  Measured at 1e5 x 90 x 5 (tools/sh_clear_time.py; the full-plane kernel on the same cloud-free
// scene: 1.009 ms): one angle per lane 0.804 ms (what dropping the cloud arithmetic and nine planes gives by itself), two
// 0.667 (198 -> 256 VGPRs, 6 spilled, still two waves per SIMD), three 0.80 (100 VGPRs spilled to scratch), five 1.52 --
// or 0.80 in 352 registers with one wave per SIMD (1 563 waves on 1 024 SIMDs: two rounds).  The per-angle sweep state in
// LDS instead (11 doubles per angle, a plain loop over the angles in 232 VGPRs): 0.884 / 0.735 / 0.737 for one / two /
// three angles per lane -- the LDS round trips cost more than the third angle's share of the common half saves.
// From the two register timings: ~190 instructions per layer shared, ~275 per angle.
// The angle's three reciprocals from one Newton core (prefix products, as modes_sh4_batched): 0.629 ms at 1e5 columns but
// 0.148 instead of 0.135 at 12 500, where one wave per SIMD waits out the longer dependency chain; not kept.
// The beam exponential exp(-tau[i+1]/u0) as the product of the two at hand (symmetric geometry, no clip): 0.641 ms against
// 0.629 with its own polynomial -- the wave-uniform test costs more than the exponential (as in k_sh, PZ_SH_OPT_EXP).
constexpr int SHC_MAX_PER_LANE = 2;

template <int NA>
__global__ __launch_bounds__(256, 2) void k_sh4_clear(const SHCArgs a)
{
    constexpr int NB = 2;
    const long w = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.nwno) return;
    const int n = a.nlayer;
    const long pitch = a.pitch;
    const int k0 = blockIdx.y * NA;
    const int na = min(NA, a.na - k0);
    const SHCArgs::Angle *const ang = a.ang + k0;
    Exp2Coef K;
    K.load();
    const double F = a.F0PI[w], rs = a.surf_reflect[w];
    // per angle: transmission to the top T, TOA functional J = kappa + zeta . v, sweep right-hand side delta, the
    // beam exponential at the layer top and the previous layer's particular solution at its bottom
    double T[NA], kappa[NA], zeta[NA][NB], delta[NA][NB], e_top[NA], p_zmn_up[NA][NB], p_zpl_up[NA][NB];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        T[k] = 1.0;
        kappa[k] = 0.0;
        e_top[k] = 1.0;                                                      // tau[0] = 0 (optics.py cumsum)
#pragma unroll
        for (int r = 0; r < NB; ++r) zeta[k][r] = delta[k][r] = p_zmn_up[k][r] = p_zpl_up[k][r] = 0.0;
    }
    Blk<NB> R, pMn, pPl, pME, pPE;
    double tau_t = 0.0;

    const int nsweep = a.state ? a.nstop : n;
    for (int i = 0; i < nsweep; ++i) {
        const long o = (long)i * pitch + w;
        const double dt = a.dtau[o], w0 = a.w0[o];
        const double tau_b = tau_t + dt;
        // ---- stream coefficients and modes (:2858-2860, 3388-3434) ----
        const double al[4] = {1.0 - w0, 3.0, fma(-w0, 0.5, 5.0), 7.0};
        ModesT4 M;
        modes_sh4x(al, dt, M, K);
        const Blk<NB> ME = scale_cols(M.Mn, M.E), PE = scale_cols(M.Pl, M.E);
        // beam source moments b_l = F w0 w_single_l P_l(-u0) / 4 pi: b_0 for all angles, b_2 = fw2 P_2(-u0) / 4 pi
        const double b0 = (F * w0) * (0.25 / PI), fw2 = F * (w0 * 0.5);
        const double sgl = div_const(w0 * F, FOURPI, R4PI) * a.psing;         // :2959-2965
        const double a0 = al[0], a2 = al[2];
        const double A_ = 3.0 * b0, a3b0 = 7.0 * b0;

        // ---- elimination, matrix half (angle independent) ----
        Blk<NB> Rn, Sm, A2i, G, Ki, Mni;
        if (i == 0) {
            Mni = invx(M.Mn);
            Rn = mmx(Mni, PE);
            Sm = Rn; A2i = Rn; G = Rn; Ki = Rn;                              // not used for i == 0
        } else {
            const Blk<NB> A1 = subx(pPl, mmx(pME, R)), A2 = subx(pMn, mmx(pPE, R));
            A2i = invx(A2);
            G = mmx(A1, A2i);
            Ki = invx(subx(M.Mn, mmx(G, M.Pl)));
            const Blk<NB> GM = subx(PE, mmx(G, ME));
            Rn = mmx(Ki, GM);
            Sm = mmx(A2i, subx(ME, mmx(M.Pl, Rn)));
            Mni = Rn;                                                        // not used for i > 0
        }

        // ---- per angle ----
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            if (k >= na) break;
            const SHCArgs::Angle &g = ang[k];
            const double iu1 = g.iu1, x = g.iu0, x2 = g.x2;
            const double edt = fexp2(dt * g.nl1, K);                         // exp(-dtau/u1)
            // particular solution eta_l = Delta_l / Delta (:3397-3416) with b_1 = b_3 = 0
            const double b2 = (fw2 * g.p2u0) * (0.25 / PI);
            const double iDel = frcp(9 * ((g.x4 - M.beta * x2) + M.gama));
            const double P = fma(a2, 7.0, -(9 * x2)), Q = fma(a0, 3.0, -x2);
            const double a3b2 = 7.0 * b2, b0x = b0 * x;
            const double B_ = a3b2 - 2 * a3b0, C_ = -b0x, E_ = -(3 * (b2 * x)), F_ = -(3 * b0x);
            double eta[4];
            eta[0] = fma(A_, P, 2 * (B_ * x2)) * iDel;
            eta[1] = fma(C_, P, -((2 * a0) * (a3b2 * x))) * iDel;
            eta[2] = fma(a3b2, Q, -(14.0 * (C_ * x))) * iDel;
            eta[3] = fma(E_, Q, 2 * (F_ * x2)) * iDel;
            const double h0 = 0.5 * eta[0], e58 = 0.625 * eta[2], m0 = -0.125 * eta[0];
            const double zpl0 = ((h0 + eta[1]) + e58) * (2 * PI), zmn0 = ((h0 - eta[1]) + e58) * (2 * PI);
            const double zpl1 = ((m0 + e58) + eta[3]) * (2 * PI), zmn1 = ((m0 + e58) - eta[3]) * (2 * PI);
            // exp(-tau/u0) at the layer top and bottom.  e_top carries the UNCLIPPED value: in the symmetric geometry
            // (u0 == u1) the bottom one is the top one times exp(-dtau/u1), already at hand -- one multiply instead of a
            // 15-instruction polynomial -- and the reference's exp(-clip35(tau/u0)) (:3418-3421) is max(., e^-35) of it
            // (exp is monotonic).  Other geometries form the exponential of the running sum tau directly.
            const double pt = e_top[k];
            const double pb = g.sym ? pt * edt : fexp2(tau_b * g.nl0, K);
            e_top[k] = pb;
            const double ed = fmax(pt, EXP_M35), eu = fmax(pb, EXP_M35);
            const double zmn_dn[NB] = {zmn0 * ed, zmn1 * ed}, zpl_dn[NB] = {zpl0 * ed, zpl1 * ed};
            const double zmn_up[NB] = {zmn0 * eu, zmn1 * eu}, zpl_up[NB] = {zpl0 * eu, zpl1 * eu};

            // source-function weights along u1 (:2898-2970, 3601-3605): the odd moments vanish
            const double e01 = fma(g.hp2u1, M.Q[0], 1.0), e23 = fma(g.hp2u1, M.Q[1], 1.0);
            const double cm[4] = {e01, e01, e23, e23};
            const double Tk = T[k];
            const double Ti = Tk * iu1, tw = Ti * w0;
            const bool noclip_lane = (iu1 + M.lam[0]) * dt <= 35.0;          // per lane; the work is wave-uniform (k_sh)
            const bool noclip = __all(noclip_lane);
            double gd[NB], gv[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const double alpha = iu1 + M.lam[r], beta = iu1 - M.lam[r];
                const double rab = frcp(alpha * beta);
                double ea = edt * M.E[r], eb = edt * M.iE[r];
                if (!noclip) {
                    const double ead = fexpk(-clip35(alpha * dt), K), ebd = fexpk(-clip35(beta * dt), K);
                    ea = noclip_lane ? ea : ead;
                    eb = noclip_lane ? eb : ebd;
                }
                const double ha = (1 - ea) * (rab * beta), hb = (1 - eb) * (rab * alpha);
                gd[r] = (tw * cm[2 * r]) * ha;
                gv[r] = ((tw * cm[2 * r + 1]) * hb) * M.E[r];
            }
            // (1 - exp(-clip35(mus dtau)))/mus (:2901-2905); mus = 2/u1 in the symmetric geometry
            const bool sq_lane = g.sym && (g.mus * dt <= 35.0);
            double e_mus = edt * edt;
            if (!__all(sq_lane)) {
                const double d = fexp2_clip(dt * g.nlm, K);
                e_mus = sq_lane ? e_mus : d;
            }
            const double om = 1 - e_mus;
            const double expon1 = (om * g.imus) * ed;
            const double Nsum = fma(g.hp2u1, eta[2], eta[0]) * expon1;      // :2919-2920, 2945-2948
            // exp(-tau_og/u0) at the layer top, unclipped (:2962): tau_og = tau without cloud
            const double e_tauo = pt;
            const double single = ((sgl * om) * e_tauo) * g.imus;
            double c = Ti * fma(w0, Nsum, single);
            const double Tn = Tk * edt;
            T[k] = Tn;
            if (i == n - 1) {             // xint[n] = flux_bot/pi = (Pl E d + Mn v + zpl_up)[0]/pi  (:2891, :2967)
                const double tp = Tn * (1.0 / PI);
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    gd[r] = fma(tp, PE.m[0][r], gd[r]);
                    gv[r] = fma(tp, M.Mn.m[0][r], gv[r]);
                }
                c = fma(tp, zpl_up[0], c);
            }
            // elimination, right-hand sides
            double z2[NB], deltan[NB];
            mtvx(Rn, gd, z2);
            if (i == 0) {
                const double bt[NB] = {a.b_top - zmn_dn[0], -a.b_top / 4 - zmn_dn[1]};    // :3479-3480
                mvx(Mni, bt, deltan);
                kappa[k] = c + dotx(gd, deltan);
#pragma unroll
                for (int r = 0; r < NB; ++r) zeta[k][r] = gv[r] - z2[r];
            } else {
                double cP[NB], cM[NB], t1[NB], t2[NB], rhs[NB], tv[NB], t[NB], z1[NB];
                mvx(pPE, delta[k], t1);
                mvx(pME, delta[k], t2);
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    cP[r] = (zpl_dn[r] - p_zpl_up[k][r]) - t1[r];
                    cM[r] = (t2[r] + p_zmn_up[k][r]) - zmn_dn[r];
                }
                mvx(G, cP, rhs);
#pragma unroll
                for (int r = 0; r < NB; ++r) rhs[r] += cM[r];
                mvx(Ki, rhs, deltan);
                mvx(M.Pl, deltan, tv);
#pragma unroll
                for (int r = 0; r < NB; ++r) tv[r] += cP[r];
                mvx(A2i, tv, t);
                kappa[k] = ((kappa[k] + dotx(zeta[k], t)) + dotx(gd, deltan)) + c;
                mtvx(Sm, zeta[k], z1);
#pragma unroll
                for (int r = 0; r < NB; ++r) zeta[k][r] = (z1[r] + gv[r]) - z2[r];
            }
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                delta[k][r] = deltan[r];
                p_zmn_up[k][r] = zmn_up[r];
                p_zpl_up[k][r] = zpl_up[r];
            }
        }
        R = Rn;
        pMn = M.Mn; pPl = M.Pl; pME = ME; pPE = PE;
        tau_t = tau_b;
    }
    if (a.state) {
        // hand the sweep over at the top of layer nstop: shared blocks (every angle chunk holds the same values; the
        // first one writes them), then this lane's angles
        double *sp = a.state + w;
        auto st = [&](int sl, double v) { sp[(long)sl * a.nwno] = v; };
        if (blockIdx.y == 0) {
            int sl = 0;
            const Blk<NB> *const blks[5] = {&R, &pMn, &pPl, &pME, &pPE};
#pragma unroll
            for (int b = 0; b < 5; ++b)
#pragma unroll
                for (int r = 0; r < NB; ++r)
#pragma unroll
                    for (int c2 = 0; c2 < NB; ++c2) st(sl++, blks[b]->m[r][c2]);
        }
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            if (k >= na) break;
            int sl = SHC_SHARED + SHC_STATE * (a.first_angle + k0 + k);
            st(sl++, T[k]);
            st(sl++, kappa[k]);
            st(sl++, e_top[k]);
#pragma unroll
            for (int r = 0; r < NB; ++r) st(sl++, zeta[k][r]);
#pragma unroll
            for (int r = 0; r < NB; ++r) st(sl++, delta[k][r]);
#pragma unroll
            for (int r = 0; r < NB; ++r) st(sl++, p_zmn_up[k][r]);
#pragma unroll
            for (int r = 0; r < NB; ++r) st(sl++, p_zpl_up[k][r]);
        }
        return;
    }
    // ---- surface rows (:3484-3494) ----
    Blk<NB> W, L;
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
        for (int s_ = 0; s_ < NB; ++s_) {
            W.m[r][s_] = fma(-rs, pME.m[r][s_], pPE.m[r][s_]);
            L.m[r][s_] = fma(-rs, pPl.m[r][s_], pMn.m[r][s_]);
        }
    const Blk<NB> Li = invx(subx(L, mmx(W, R)));
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        if (k >= na) break;
        const SHCArgs::Angle &g = ang[k];
        const double bsf = (rs * g.u0) * F * fexpk(-(tau_t * g.iu0), K);     // :2863-2864
        const double bs[NB] = {bsf, -bsf / 4};
        double wd[NB], rhs[NB], v[NB];
        mvx(W, delta[k], wd);
#pragma unroll
        for (int r = 0; r < NB; ++r) rhs[r] = ((bs[r] - p_zpl_up[k][r]) + rs * p_zmn_up[k][r]) - wd[r];
        mvx(Li, rhs, v);
        a.xint[(long)(k0 + k) * a.nwno + w] = kappa[k] + dotx(zeta[k], v);
    }
}
#pragma clang fp contract(fast)

static int launch_sh4_clear(picaso_ctx *ctx, SHCArgs &a, int nang, const double *ubar0, const double *ubar1, double *xint)
{
    const int m = sh_angles_per_lane(ctx, a.nwno, nang, 190.0, 275.0, "PICASO_AMD_SHC_ANGLES", SHC_MAX_PER_LANE);
    const int nchunk = (nang + m - 1) / m;
    const int per_lane = (nang + nchunk - 1) / nchunk;
    const int per_launch = (SH_MAX_ANG / per_lane) * per_lane;
    for (int done = 0; done < nang; done += per_launch) {
        const int na = (nang - done < per_launch) ? nang - done : per_launch;
        for (int k = 0; k < na; ++k) {
#pragma clang fp contract(off)
            const SHArgs::Angle s = make_sh_angle(ubar0[done + k], ubar1[done + k], false);
            SHCArgs::Angle &g = a.ang[k];
            g.u0 = s.u0; g.iu0 = s.iu0; g.iu1 = s.iu1; g.mus = s.mus; g.imus = s.imus;
            g.nl0 = s.nl0; g.nl1 = s.nl1; g.nlm = s.nlm;
            g.x2 = g.iu0 * g.iu0;
            g.x4 = g.x2 * g.x2;
            const double m0 = -s.u0, m1 = s.u1;                                  // legP (fluxes.py:3639-3646)
            g.p2u0 = (3 * m0 * m0 - 1) / 2;
            g.hp2u1 = 0.5 * ((3 * m1 * m1 - 1) / 2);
            g.sym = (s.u0 == s.u1) ? 1 : 0;
        }
        a.na = na;
        a.first_angle = done;
        a.xint = xint + (size_t)done * a.nwno;
        const dim3 grid((unsigned)((a.nwno + 255) / 256), (unsigned)((na + per_lane - 1) / per_lane));
        if (per_lane == 1) hipLaunchKernelGGL(k_sh4_clear<1>, grid, dim3(256), 0, ctx->stream, a);
        else hipLaunchKernelGGL(k_sh4_clear<2>, grid, dim3(256), 0, ctx->stream, a);
        PZ_HIP(ctx, hipGetLastError());
    }
    return 0;
}

constexpr int SH_TOP_MIN = 4;     // cloud-free layers from which the split (two launches, a state hand-over) is taken

// PICASO_AMD_SH_CHECK_TOP=1: count the elements of the first `top` layers that contradict the caller's statement
// (picaso_get_reflected_SH_top_dev) -- a synchronising debug aid, the tests run with it.
__global__ void k_sh_check_top(const SHArgs a, int top, unsigned long long *bad)
{
#pragma clang fp contract(off)
    const long w = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.nwno) return;
    unsigned n = 0;
    for (int i = 0; i < top; ++i) {
        const long o = (long)i * a.pitch + w;
        const double dt = a.dtau[o];
        n += (a.ftau_cld[o] != 0.0) + (a.cosb_og[o] != 0.0) + (a.f_deltaM[o] != 0.0) + (a.ftau_ray[o] != 1.0) +
             (a.dtau_og[o] != dt) + (a.w0_og[o] != a.w0[o]);
        if (a.tau && a.tau_og)                  // (left out: re-derived as running sums in the kernels)
            n += (a.tau[o + a.pitch] != a.tau[o] + dt) + (a.tau_og[o] != a.tau[o]) + (i == 0 && a.tau[o] != 0.0);
    }
    if (n) atomicAdd(bad, (unsigned long long)n);
}

static int sh_check_top(picaso_ctx *ctx, const SHArgs &a, int top)
{
    PZ_TRY(ck_scratch_reserve(ctx, 256));
    unsigned long long *d = (unsigned long long *)ctx->ck_scratch, h = 0;
    PZ_HIP(ctx, hipMemsetAsync(d, 0, sizeof(h), ctx->stream));
    hipLaunchKernelGGL(k_sh_check_top, dim3((unsigned)((a.nwno + 255) / 256)), dim3(256), 0, ctx->stream, a, top, d);
    PZ_HIP(ctx, hipGetLastError());
    PZ_HIP(ctx, hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h)
        return fail(ctx, "get_reflected_SH: cloud_free_above = %d, but %llu plane elements of those layers are not what a "
                         "cloud-free layer holds (ftau_cld = cosb_og = f_deltaM = 0, ftau_ray = 1, dtau_og = dtau, w0_og = w0, "
                         "tau = tau_og = the running sum of dtau)", top, h);
    return 0;
}

static int launch_sh(picaso_ctx *ctx, SHArgs &a, int nang, bool thermal)
{
    const int block = 256;
    const unsigned ncg = (unsigned)((a.nwno + block - 1) / block);          // column groups
    a.ncg = ncg;
    a.xcd_order = (ncg * (unsigned)nang >= 4u * (unsigned)ctx->ncu) && !getenv("PICASO_AMD_SH_ANGLE_MAJOR");
    const unsigned nspec = a.batch ? (unsigned)a.nspec_ : 1u;
    if (nspec > 1) a.xcd_order = (ncg * (unsigned)nang * nspec >= 4u * (unsigned)ctx->ncu) && !getenv("PICASO_AMD_SH_ANGLE_MAJOR");
    const dim3 grid(a.xcd_order ? ((ncg + 7u) / 8u) * 8u * (unsigned)nang : ncg * (unsigned)nang, a.batch ? nspec : 1u);
    const bool flx = a.flux != nullptr;
    const bool fast = !thermal && !flx && !getenv("PICASO_AMD_SH_GENERIC") && a.w_single_form == 0 &&
                      a.w_multi_form == 0 && a.psingle_form == 0 && a.single_form == 0 && a.w_single_rayleigh == 1 &&
                      a.w_multi_rayleigh == 1 && a.psingle_rayleigh == 1 && a.frac_c == 2.0 &&   // config.json defaults
                      (double)a.pitch * (a.nlayer + 1) * 8.0 < 4294967296.0;                   // 32-bit plane offsets
    if (a.batch && (thermal || flx)) return fail(ctx, "SH batch: reflected light without layer fluxes only");
    const bool drvt = !thermal && (!a.tau || !a.tau_og);   // level planes left out: picaso_reflected_SH_can_derive_levels
    if (drvt && (!fast || a.tau || a.tau_og))
        return fail(ctx, "get_reflected_SH: tau and tau_og may be left out (both) only with the reference's default "
                         "options, flx = 0 and planes below 4 GB (picaso_reflected_SH_can_derive_levels)");
    if (!thermal && !flx) {                         // single spectrum (a.batch = NULL) or a batch: the same kernel
#define PZ_GO(KERNEL) hipLaunchKernelGGL(KERNEL, grid, dim3(block), 0, ctx->stream, a)
        if (a.stream == 4) {
            if (drvt) PZ_GO((k_sh_refl<2, true, true>));
            else if (fast) PZ_GO((k_sh_refl<2, true>));
            else PZ_GO((k_sh_refl<2, false>));
        } else {
            if (drvt) PZ_GO((k_sh_refl<1, true, true>));
            else if (fast) PZ_GO((k_sh_refl<1, true>));
            else PZ_GO((k_sh_refl<1, false>));
        }
#undef PZ_GO
        PZ_HIP(ctx, hipGetLastError());
        return 0;
    }
    if (a.stream == 4) {
        if (thermal) hipLaunchKernelGGL((k_sh<2, true, false>), grid, dim3(block), 0, ctx->stream, a);
        else hipLaunchKernelGGL((k_sh<2, false, true>), grid, dim3(block), 0, ctx->stream, a);
    } else {
        if (thermal) hipLaunchKernelGGL((k_sh<1, true, false>), grid, dim3(block), 0, ctx->stream, a);
        else hipLaunchKernelGGL((k_sh<1, false, true>), grid, dim3(block), 0, ctx->stream, a);
    }
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

// All angles of a call go into one launch (grid.y = angle, SH_MAX_ANG per launch): a 1e5-column
// spectrum is only 1.5 waves per SIMD per angle, five angles together fill the chip evenly.
static SHArgs::Angle make_sh_angle(double u0, double u1, bool thermal)
{
    SHArgs::Angle g;
    g.u1 = u1;
    g.iu1 = 1.0 / g.u1;
    g.nl1 = -LOG2E * g.iu1;
    if (thermal) {
        g.u0 = g.iu0 = g.mus = g.imus = g.nl0 = g.nlm = 0.0;
    } else {
        g.u0 = u0;
        g.iu0 = 1.0 / g.u0;
        g.mus = (g.u1 + g.u0) / (g.u1 * g.u0);                       // fluxes.py:2899
        g.imus = 1.0 / g.mus;
        g.nl0 = -LOG2E * g.iu0;
        g.nlm = -LOG2E * g.mus;
    }
    return g;
}

static int launch_sh_angles(picaso_ctx *ctx, SHArgs &a, int nang, const double *ubar0, const double *ubar1,
                            double *xint_at_top, double *flux, bool thermal)
{
    const int nb = a.stream / 2, nslot = 4 * nb * nb + 5 * nb;
    const size_t per_angle = (size_t)nslot * a.nlayer * a.nwno;
    // flx = 1 keeps the sweep state of every layer: limit the angles per launch to ~8 GB of scratch
    int max_ang = SH_MAX_ANG;
    if (flux) {
        const size_t cap = (size_t)8 << 30;
        max_ang = (int)(cap / (per_angle * sizeof(double)));
        if (max_ang < 1) max_ang = 1;
        if (max_ang > SH_MAX_ANG) max_ang = SH_MAX_ANG;
        PZ_TRY(ck_scratch_reserve(ctx, sizeof(double) * per_angle * (size_t)(nang < max_ang ? nang : max_ang)));
    }
    for (int done = 0; done < nang; done += max_ang) {
        const int m = (nang - done < max_ang) ? nang - done : max_ang;
        a.flux = flux ? flux + (size_t)done * a.stream * (a.nlayer + 1) * a.nwno : nullptr;
        a.scratch = flux ? ctx->ck_scratch : nullptr;
        for (int k = 0; k < m; ++k) a.ang[k] = make_sh_angle(thermal ? 0.0 : ubar0[done + k], ubar1[done + k], thermal);
        a.first_angle = done;
        a.nang = m;
        a.xint = xint_at_top + (size_t)done * a.nwno;
        PZ_TRY(launch_sh(ctx, a, m, thermal));
    }
    return 0;
}

}  // namespace pz

using namespace pz;

extern "C" {

int picaso_reflected_SH_can_derive(int stream, int w_single_form, int w_multi_form, int psingle_form,
                                   int w_single_rayleigh, int w_multi_rayleigh, int psingle_rayleigh, double frac_c,
                                   int single_form, int flx)
{
    if (getenv("PICASO_AMD_SH_NO_CLEAR")) return 0;
    return stream == 4 && !flx && w_single_form == 0 && w_multi_form == 0 && psingle_form == 0 && single_form == 0 &&
           w_single_rayleigh == 1 && w_multi_rayleigh == 1 && psingle_rayleigh == 1 && frac_c == 2.0;
}

int picaso_reflected_SH_can_derive_levels(int nlevel, long plane_pitch, int stream, int w_single_form, int w_multi_form,
                                          int psingle_form, int w_single_rayleigh, int w_multi_rayleigh,
                                          int psingle_rayleigh, double frac_c, int single_form, int flx)
{
    if (getenv("PICASO_AMD_SH_GENERIC") || getenv("PICASO_AMD_SH_ALL_PLANES")) return 0;
    return (stream == 2 || stream == 4) && !flx && w_single_form == 0 && w_multi_form == 0 && psingle_form == 0 &&
           single_form == 0 && w_single_rayleigh == 1 && w_multi_rayleigh == 1 && psingle_rayleigh == 1 && frac_c == 2.0 &&
           nlevel >= 2 && (double)plane_pitch * nlevel * 8.0 < 4294967296.0;
}

int picaso_get_reflected_SH_dev(picaso_ctx *ctx, int nlevel, int nwno, long plane_pitch, int numg, int numt,
                                const double *dtau, const double *tau, const double *w0,
                                const double *cosb, const double *ftau_cld, const double *ftau_ray,
                                const double *f_deltaM, const double *dtau_og, const double *tau_og,
                                const double *w0_og, const double *cosb_og, const double *surf_reflect,
                                const double *ubar0, const double *ubar1, double cos_theta,
                                const double *F0PI, int w_single_form, int w_multi_form,
                                int psingle_form, int w_single_rayleigh, int w_multi_rayleigh,
                                int psingle_rayleigh, double frac_a, double frac_b, double frac_c,
                                double constant_back, double constant_forward, int stream,
                                double b_top, int flx, int single_form, int compound_f_deltaM,
                                double *xint_at_top, double *flux, const double *gweight,
                                const double *tweight, double *albedo)
{
    return picaso_get_reflected_SH_top_dev(ctx, nlevel, nwno, plane_pitch, numg, numt, dtau, tau, w0, cosb, ftau_cld,
                                           ftau_ray, f_deltaM, dtau_og, tau_og, w0_og, cosb_og, surf_reflect, ubar0, ubar1,
                                           cos_theta, F0PI, w_single_form, w_multi_form, psingle_form, w_single_rayleigh,
                                           w_multi_rayleigh, psingle_rayleigh, frac_a, frac_b, frac_c, constant_back,
                                           constant_forward, stream, b_top, flx, single_form, compound_f_deltaM, 0,
                                           xint_at_top, flux, gweight, tweight, albedo);
}

int picaso_get_reflected_SH_top_dev(picaso_ctx *ctx, int nlevel, int nwno, long plane_pitch, int numg, int numt,
                                    const double *dtau, const double *tau, const double *w0,
                                    const double *cosb, const double *ftau_cld, const double *ftau_ray,
                                    const double *f_deltaM, const double *dtau_og, const double *tau_og,
                                    const double *w0_og, const double *cosb_og, const double *surf_reflect,
                                    const double *ubar0, const double *ubar1, double cos_theta,
                                    const double *F0PI, int w_single_form, int w_multi_form,
                                    int psingle_form, int w_single_rayleigh, int w_multi_rayleigh,
                                    int psingle_rayleigh, double frac_a, double frac_b, double frac_c,
                                    double constant_back, double constant_forward, int stream,
                                    double b_top, int flx, int single_form, int compound_f_deltaM,
                                    int cloud_free_above, double *xint_at_top, double *flux, const double *gweight,
                                    const double *tweight, double *albedo)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nlevel < 2 || nwno < 1 || numg < 1 || numt < 1) return fail(ctx, "get_reflected_SH: bad sizes");
    if (stream != 2 && stream != 4) return fail(ctx, "get_reflected_SH: stream must be 2 or 4, got %d", stream);
    if (flx && !flux) return fail(ctx, "get_reflected_SH: flx=1 needs the flux output (numg,numt,stream*nlevel,nwno)");
    PZ_NEED(ctx, "get_reflected_SH", surf_reflect, ubar0, ubar1, F0PI, xint_at_top);
    if (flx && plane_pitch != nwno) return fail(ctx, "get_reflected_SH: flx=1 needs contiguous planes");
    if (plane_pitch < nwno) return fail(ctx, "get_reflected_SH: plane_pitch < nwno");
    if (!dtau || !w0) return fail(ctx, "get_reflected_SH: dtau and w0 are required");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    (void)cosb;                                                           // never read (fluxes.py:2675-2976)
    const double *const derivable[8] = {tau, ftau_cld, ftau_ray, f_deltaM, dtau_og, tau_og, w0_og, cosb_og};
    int nnull = 0;
    for (int j = 0; j < 8; ++j) nnull += derivable[j] ? 0 : 1;
    // only the two level planes left out: the running-product form of the beam exponentials (k_sh<.., DRVT>)
    const bool levels_only = (nnull == 2 && !tau && !tau_og);
    if (levels_only &&
        !picaso_reflected_SH_can_derive_levels(nlevel, plane_pitch, stream, w_single_form, w_multi_form, psingle_form,
                                               w_single_rayleigh, w_multi_rayleigh, psingle_rayleigh, frac_c, single_form, flx))
        return fail(ctx, "get_reflected_SH: tau and tau_og may be left out only with the reference's default phase-function "
                         "options, flx = 0 and planes below 4 GB (picaso_reflected_SH_can_derive_levels); pass them");
    if (nnull && !levels_only) {
        // cloud-free form: dtau and w0 only, everything else known (k_sh4_clear)
        if (nnull != 8)
            return fail(ctx, "get_reflected_SH: leave out all of tau, cosb, ftau_cld, ftau_ray, f_deltaM, dtau_og, tau_og, "
                             "w0_og, cosb_og (a cloud-free column: they are constants, copies and running sums) or none");
        if (!picaso_reflected_SH_can_derive(stream, w_single_form, w_multi_form, psingle_form, w_single_rayleigh,
                                            w_multi_rayleigh, psingle_rayleigh, frac_c, single_form, flx))
            return fail(ctx, "get_reflected_SH: the cloud-free form (dtau and w0 only) is built for stream 4, the default "
                             "phase-function options and flx = 0; pass all eleven planes");
        if (!surf_reflect || !F0PI || !ubar0 || !ubar1 || !xint_at_top) return fail(ctx, "get_reflected_SH: null argument");
        SHCArgs c{};
        c.nlayer = nlevel - 1; c.nwno = nwno; c.pitch = plane_pitch;
        c.dtau = dtau; c.w0 = w0; c.surf_reflect = surf_reflect; c.F0PI = F0PI;
        {
#pragma clang fp contract(off)
            c.psing = 0.75 * (1 + cos_theta * cos_theta);
        }
        c.b_top = b_top;
        PZ_TRY(launch_sh4_clear(ctx, c, numg * numt, ubar0, ubar1, xint_at_top));
        if (albedo && gweight && tweight)
            PZ_TRY(picaso_compress_disco_dev(ctx, nwno, cos_theta, xint_at_top, gweight, numg, tweight, numt, F0PI, albedo));
        return 0;
    }
    if (cloud_free_above < 0) return fail(ctx, "get_reflected_SH: cloud_free_above must be >= 0, got %d", cloud_free_above);
    SHArgs a{};
    a.nlayer = nlevel - 1; a.nwno = nwno; a.stream = stream; a.pitch = plane_pitch;
    a.dtau = dtau; a.tau = tau; a.w0 = w0; a.ftau_cld = ftau_cld; a.ftau_ray = ftau_ray;
    a.f_deltaM = f_deltaM; a.dtau_og = dtau_og; a.tau_og = tau_og; a.w0_og = w0_og; a.cosb_og = cosb_og;
    a.surf_reflect = surf_reflect; a.F0PI = F0PI; a.cos_theta = cos_theta;
    a.w_single_form = w_single_form; a.w_multi_form = w_multi_form; a.psingle_form = psingle_form;
    a.w_single_rayleigh = w_single_rayleigh; a.w_multi_rayleigh = w_multi_rayleigh;
    a.psingle_rayleigh = psingle_rayleigh; a.single_form = single_form;
    a.frac_a = frac_a; a.frac_b = frac_b; a.frac_c = frac_c; a.constant_back = constant_back;
    a.constant_forward = constant_forward; a.b_top = b_top;
    a.compound = compound_f_deltaM ? 1 : 0;
    // A cloud-free top (the caller's statement: no cloud in layers 0 .. cloud_free_above - 1 of any column): those
    // layers go through k_sh4_clear, two angles per lane sharing the angle-independent half, which leaves the sweep
    // state at the top of the first cloudy layer for k_sh to continue from.  Worth a second launch from a few layers on.
    const int nlayer = nlevel - 1;
    const int top = cloud_free_above > nlayer ? nlayer : cloud_free_above;
    if (top >= SH_TOP_MIN && !getenv("PICASO_AMD_SH_NO_TOP") &&
        picaso_reflected_SH_can_derive(stream, w_single_form, w_multi_form, psingle_form, w_single_rayleigh,
                                       w_multi_rayleigh, psingle_rayleigh, frac_c, single_form, flx)) {
        if (!surf_reflect || !F0PI || !ubar0 || !ubar1 || !xint_at_top) return fail(ctx, "get_reflected_SH: null argument");
        if (getenv("PICASO_AMD_SH_CHECK_TOP")) PZ_TRY(sh_check_top(ctx, a, top));
        const int nang = numg * numt;
        SHCArgs c{};
        c.nlayer = nlayer; c.nwno = nwno; c.pitch = plane_pitch;
        c.dtau = dtau; c.w0 = w0; c.surf_reflect = surf_reflect; c.F0PI = F0PI;
        {
#pragma clang fp contract(off)
            c.psing = 0.75 * (1 + cos_theta * cos_theta);
        }
        c.b_top = b_top;
        if (top < nlayer) {
            PZ_TRY(ck_scratch_reserve(ctx, sizeof(double) * (size_t)(SHC_SHARED + SHC_STATE * nang) * nwno));
            c.nstop = top;
            c.state = ctx->ck_scratch;
            a.start_layer = top;
            a.state = ctx->ck_scratch;
        }
        PZ_TRY(launch_sh4_clear(ctx, c, nang, ubar0, ubar1, xint_at_top));
        if (top < nlayer) PZ_TRY(launch_sh_angles(ctx, a, nang, ubar0, ubar1, xint_at_top, nullptr, false));
    } else {
        PZ_TRY(launch_sh_angles(ctx, a, numg * numt, ubar0, ubar1, xint_at_top, flx ? flux : nullptr, false));
    }
    if (albedo && gweight && tweight)
        PZ_TRY(picaso_compress_disco_dev(ctx, nwno, cos_theta, xint_at_top, gweight, numg, tweight, numt, F0PI, albedo));
    return 0;
}

int picaso_get_reflected_SH_batch_dev(picaso_ctx *ctx, int nspec, int nlevel, int nwno, long plane_pitch, int numg,
                                      int numt, const double *const *dtau, const double *const *tau,
                                      const double *const *w0, const double *const *cosb,
                                      const double *const *ftau_cld, const double *const *ftau_ray,
                                      const double *const *f_deltaM, const double *const *dtau_og,
                                      const double *const *tau_og, const double *const *w0_og,
                                      const double *const *cosb_og, const double *const *surf_reflect, int ngeom,
                                      const double *ubar0, const double *ubar1, const double *cos_theta,
                                      const double *const *F0PI, int w_single_form, int w_multi_form,
                                      int psingle_form, int w_single_rayleigh, int w_multi_rayleigh,
                                      int psingle_rayleigh, double frac_a, double frac_b, double frac_c,
                                      double constant_back, double constant_forward, int stream, double b_top,
                                      int single_form, int compound_f_deltaM, double *const *xint_at_top,
                                      const double *gweight, const double *tweight, double *const *albedo)
{
    (void)cosb;
    if (!ctx) return fail(nullptr, "null context");
    if (nspec < 1) return fail(ctx, "get_reflected_SH_batch: nspec must be >= 1, got %d", nspec);
    if (nlevel < 2 || nwno < 1 || numg < 1 || numt < 1) return fail(ctx, "get_reflected_SH_batch: bad sizes");
    if (stream != 2 && stream != 4) return fail(ctx, "get_reflected_SH_batch: stream must be 2 or 4, got %d", stream);
    if (plane_pitch < nwno) return fail(ctx, "get_reflected_SH_batch: plane_pitch < nwno");
    if (ngeom != 1 && ngeom != nspec)
        return fail(ctx, "get_reflected_SH_batch: ngeom must be 1 (one geometry for all) or nspec, got %d", ngeom);
    const int nang = numg * numt;
    if (nang > SH_MAX_ANG) return fail(ctx, "get_reflected_SH_batch: at most %d disk angles", SH_MAX_ANG);
    const double *const *pl[10] = {dtau, tau, w0, ftau_cld, ftau_ray, f_deltaM, dtau_og, tau_og, w0_og, cosb_og};
    for (int j = 0; j < 10; ++j) {
        if (!pl[j]) return fail(ctx, "get_reflected_SH_batch: null plane pointer array");
        for (int s = 0; s < nspec; ++s)
            if (!pl[j][s]) return fail(ctx, "get_reflected_SH_batch: plane %d of spectrum %d is NULL", j, s);
    }
    if (!surf_reflect || !F0PI || !xint_at_top || !ubar0 || !ubar1 || !cos_theta)
        return fail(ctx, "get_reflected_SH_batch: null argument");
    const bool fuse = albedo && gweight && tweight;
    for (int s = 0; s < nspec; ++s)
        if (!surf_reflect[s] || !F0PI[s] || !xint_at_top[s] || (fuse && !albedo[s]))
            return fail(ctx, "get_reflected_SH_batch: null per-spectrum pointer (spectrum %d)", s);
    if (sizeof(SHBatchItem) * (size_t)nspec > picaso_ctx::SLOT_BYTES)
        return fail(ctx, "get_reflected_SH_batch: at most %zu spectra per call", picaso_ctx::SLOT_BYTES / sizeof(SHBatchItem));
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<SHBatchItem> items((size_t)nspec);
    for (int s = 0; s < nspec; ++s) {
        SHBatchItem &it = items[(size_t)s];
        it.dtau = dtau[s]; it.tau = tau[s]; it.w0 = w0[s]; it.ftau_cld = ftau_cld[s]; it.ftau_ray = ftau_ray[s];
        it.f_deltaM = f_deltaM[s]; it.dtau_og = dtau_og[s]; it.tau_og = tau_og[s]; it.w0_og = w0_og[s];
        it.cosb_og = cosb_og[s]; it.surf_reflect = surf_reflect[s]; it.F0PI = F0PI[s]; it.xint = xint_at_top[s];
        it.cos_theta = cos_theta[ngeom > 1 ? s : 0];
        const double *u0 = ubar0 + (ngeom > 1 ? (size_t)s * nang : 0), *u1 = ubar1 + (ngeom > 1 ? (size_t)s * nang : 0);
        for (int k = 0; k < nang; ++k) it.ang[k] = make_sh_angle(u0[k], u1[k], false);
    }
    const void *d = nullptr;
    PZ_TRY(table_upload(ctx, items.data(), sizeof(SHBatchItem) * items.size(), &d));
    SHArgs a{};
    a.nlayer = nlevel - 1; a.nwno = nwno; a.stream = stream; a.pitch = plane_pitch;
    a.dtau = dtau[0]; a.tau = tau[0]; a.w0 = w0[0]; a.ftau_cld = ftau_cld[0]; a.ftau_ray = ftau_ray[0];
    a.f_deltaM = f_deltaM[0]; a.dtau_og = dtau_og[0]; a.tau_og = tau_og[0]; a.w0_og = w0_og[0]; a.cosb_og = cosb_og[0];
    a.surf_reflect = surf_reflect[0]; a.F0PI = F0PI[0]; a.cos_theta = cos_theta[0];
    a.w_single_form = w_single_form; a.w_multi_form = w_multi_form; a.psingle_form = psingle_form;
    a.w_single_rayleigh = w_single_rayleigh; a.w_multi_rayleigh = w_multi_rayleigh;
    a.psingle_rayleigh = psingle_rayleigh; a.single_form = single_form;
    a.frac_a = frac_a; a.frac_b = frac_b; a.frac_c = frac_c; a.constant_back = constant_back;
    a.constant_forward = constant_forward; a.b_top = b_top;
    a.compound = compound_f_deltaM ? 1 : 0;
    a.batch = (const SHBatchItem *)d;
    a.nspec_ = nspec;
    a.first_angle = 0;
    a.nang = nang;
    a.xint = xint_at_top[0];
    a.flux = nullptr;
    a.scratch = nullptr;
    PZ_TRY(launch_sh(ctx, a, nang, false));
    if (fuse)
        for (int s = 0; s < nspec; ++s)
            PZ_TRY(picaso_compress_disco_dev(ctx, nwno, cos_theta[ngeom > 1 ? s : 0], xint_at_top[s], gweight, numg,
                                             tweight, numt, F0PI[s], albedo[s]));
    return 0;
}

int picaso_get_reflected_SH(picaso_ctx *ctx, int nlevel, int nwno, int numg, int numt, const double *dtau,
                            const double *tau, const double *w0, const double *cosb,
                            const double *ftau_cld, const double *ftau_ray, const double *f_deltaM,
                            const double *dtau_og, const double *tau_og, const double *w0_og,
                            const double *cosb_og, const double *surf_reflect, const double *ubar0,
                            const double *ubar1, double cos_theta, const double *F0PI,
                            int w_single_form, int w_multi_form, int psingle_form,
                            int w_single_rayleigh, int w_multi_rayleigh, int psingle_rayleigh,
                            double frac_a, double frac_b, double frac_c, double constant_back,
                            double constant_forward, int stream, double b_top, int flx,
                            int single_form, int compound_f_deltaM, double *xint_at_top, double *flux)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nlevel < 2 || nwno < 1 || numg < 1 || numt < 1) return fail(ctx, "get_reflected_SH: bad sizes");
    if (stream != 2 && stream != 4) return fail(ctx, "get_reflected_SH: stream must be 2 or 4, got %d", stream);
    if (flx && !flux) return fail(ctx, "get_reflected_SH: flx=1 needs the flux output (numg,numt,stream*nlevel,nwno)");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nl = (size_t)(nlevel - 1) * nwno, nv = (size_t)nlevel * nwno, nang = (size_t)numg * numt;
    const size_t nflux = flx ? nang * stream * nlevel * nwno : 0;
    PZ_TRY(arena_reset(ctx, sizeof(double) * (9 * nl + 2 * nv + 2 * (size_t)nwno + nang * nwno + nflux) + 64 * 256));
    const double *d[10], *d_rs, *d_f0;
    const double *h[10] = {dtau, tau, w0, ftau_cld, ftau_ray, f_deltaM, dtau_og, tau_og, w0_og, cosb_og};
    for (int j = 0; j < 10; ++j) {                    // planes left out (cloud-free form) stay NULL for the _dev entry
        d[j] = nullptr;
        if (h[j]) PZ_TRY(arena_upload(ctx, h[j], (j == 1 || j == 7) ? nv : nl, &d[j]));
    }
    PZ_TRY(arena_upload(ctx, surf_reflect, (size_t)nwno, &d_rs));
    PZ_TRY(arena_upload(ctx, F0PI, (size_t)nwno, &d_f0));
    double *d_x = (double *)arena_take(ctx, sizeof(double) * nang * nwno);
    double *d_fl = nflux ? (double *)arena_take(ctx, sizeof(double) * nflux) : nullptr;
    if (!d_x || (nflux && !d_fl)) return fail(ctx, "arena exhausted");
    PZ_TRY(picaso_get_reflected_SH_dev(ctx, nlevel, nwno, nwno, numg, numt, d[0], d[1], d[2], cosb, d[3], d[4], d[5],
                                       d[6], d[7], d[8], d[9], d_rs, ubar0, ubar1, cos_theta, d_f0, w_single_form,
                                       w_multi_form, psingle_form, w_single_rayleigh, w_multi_rayleigh,
                                       psingle_rayleigh, frac_a, frac_b, frac_c, constant_back, constant_forward,
                                       stream, b_top, flx, single_form, compound_f_deltaM, d_x, d_fl, nullptr,
                                       nullptr, nullptr));
    PZ_HIP(ctx, hipMemcpyAsync(xint_at_top, d_x, sizeof(double) * nang * nwno, hipMemcpyDeviceToHost, ctx->stream));
    if (nflux) PZ_HIP(ctx, hipMemcpyAsync(flux, d_fl, sizeof(double) * nflux, hipMemcpyDeviceToHost, ctx->stream));
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int picaso_get_thermal_SH_dev(picaso_ctx *ctx, int nlevel, const double *wno, int nwno, long plane_pitch,
                              int numg, int numt, const double *tlevel, const double *dtau,
                              const double *tau, const double *w0, const double *cosb_og,
                              const double *plevel, const double *ubar1, const double *surf_reflect,
                              int stream, int hard_surface, int cosb_differs_from_cosb_og, int flx,
                              double *xint_at_top, const double *gweight, const double *tweight,
                              double *flux_disk)
{
    (void)tau;
    if (!ctx) return fail(nullptr, "null context");
    if (nlevel < 2 || nwno < 1 || numg < 1 || numt < 1) return fail(ctx, "get_thermal_SH: bad sizes");
    PZ_NEED(ctx, "get_thermal_SH", wno, tlevel, dtau, w0, cosb_og, plevel, ubar1, surf_reflect, xint_at_top);
    if (stream != 2 && stream != 4) return fail(ctx, "get_thermal_SH: stream must be 2 or 4, got %d", stream);
    if (flx) return fail(ctx, "get_thermal_SH: flx=1 is broken in the reference (fluxes.py:3102) and not built");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<double> tab(2 * (size_t)nlevel);
    for (int i = 0; i < nlevel; ++i) { tab[i] = tlevel[i]; tab[nlevel + i] = plevel[i]; }
    const void *d_tab = nullptr;
    PZ_TRY(table_upload(ctx, tab.data(), sizeof(double) * tab.size(), &d_tab));
    SHArgs a{};
    a.nlayer = nlevel - 1; a.nwno = nwno; a.stream = stream; a.pitch = plane_pitch;
    a.dtau = dtau; a.w0 = w0; a.cosb_og = cosb_og; a.surf_reflect = surf_reflect;
    a.wno = wno; a.tlevel = (const double *)d_tab; a.plevel = a.tlevel + nlevel;
    a.hard_surface = hard_surface; a.use_ff = cosb_differs_from_cosb_og;
    if (getenv("PICASO_AMD_SH_THERMAL_PER_ANGLE")) {            // the round-2 kernel: one wave per angle, nothing shared
        PZ_TRY(launch_sh_angles(ctx, a, numg * numt, nullptr, ubar1, xint_at_top, nullptr, true));
    } else {
        SHTArgs t{};
        t.nlayer = nlevel - 1; t.nwno = nwno; t.pitch = plane_pitch;
        t.dtau = dtau; t.w0 = w0; t.cosb_og = cosb_og; t.surf_reflect = surf_reflect; t.wno = wno;
        t.tlevel = a.tlevel; t.plevel = a.plevel; t.hard_surface = hard_surface; t.use_ff = cosb_differs_from_cosb_og;
        PZ_TRY(launch_sh_thermal(ctx, t, stream, numg * numt, ubar1, xint_at_top));
    }
    if (flux_disk && gweight && tweight)
        PZ_TRY(picaso_compress_thermal_dev(ctx, (size_t)nwno, xint_at_top, gweight, numg, tweight, numt, flux_disk));
    return 0;
}

int picaso_get_thermal_SH(picaso_ctx *ctx, int nlevel, const double *wno, int nwno, int numg, int numt,
                          const double *tlevel, const double *dtau, const double *tau, const double *w0,
                          const double *cosb_og, const double *plevel, const double *ubar1,
                          const double *surf_reflect, int stream, int hard_surface,
                          int cosb_differs_from_cosb_og, int flx, double *xint_at_top)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nlevel < 2 || nwno < 1 || numg < 1 || numt < 1) return fail(ctx, "get_thermal_SH: bad sizes");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nl = (size_t)(nlevel - 1) * nwno, nang = (size_t)numg * numt;
    PZ_TRY(arena_reset(ctx, sizeof(double) * (3 * nl + 2 * (size_t)nwno + nang * nwno) + 32 * 256));
    const double *d_dtau, *d_w0, *d_cbo, *d_rs, *d_wno;
    PZ_TRY(arena_upload(ctx, dtau, nl, &d_dtau));
    PZ_TRY(arena_upload(ctx, w0, nl, &d_w0));
    PZ_TRY(arena_upload(ctx, cosb_og, nl, &d_cbo));
    PZ_TRY(arena_upload(ctx, surf_reflect, (size_t)nwno, &d_rs));
    PZ_TRY(arena_upload(ctx, wno, (size_t)nwno, &d_wno));
    double *d_x = (double *)arena_take(ctx, sizeof(double) * nang * nwno);
    if (!d_x) return fail(ctx, "arena exhausted");
    PZ_TRY(picaso_get_thermal_SH_dev(ctx, nlevel, d_wno, nwno, nwno, numg, numt, tlevel, d_dtau, tau, d_w0, d_cbo,
                                     plevel, ubar1, d_rs, stream, hard_surface, cosb_differs_from_cosb_og, flx, d_x,
                                     nullptr, nullptr, nullptr));
    PZ_HIP(ctx, hipMemcpyAsync(xint_at_top, d_x, sizeof(double) * nang * nwno, hipMemcpyDeviceToHost, ctx->stream));
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
