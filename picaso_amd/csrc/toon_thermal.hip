// Toon89 two-stream thermal emission, top-of-atmosphere flux -- gfx950.
//
// Replaces fluxes.get_thermal_1d / get_thermal_3d (reference picaso/fluxes.py:1682-1912,
// :2147-2352) for the spectrum path: flux_at_top = flux_plus_mdpt[:, :, 0, :] (fluxes.py:1910).
//
// Same lane mapping and the same single top-down sweep as toon_reflected.hip: the two-stream
// system (hemispheric mean, linear-in-tau Planck source) has ONE right-hand side per wavelength,
// so (rho, delta) are shared by all angles; only the TOA functional (kappa, zeta) is per angle.
// The Planck function is evaluated on the fly per level (never materialised, fluxes.py:1752).
// The upward source-function recursion (Toon Table 3, fluxes.py:1897-1907)
//     F+[i] = F+[i+1] e_i + G_i/(lam mu - 1)(EP e - 1) + H_i/(lam mu + 1)(1 - EM e) + alpha terms,
//     F+m[0] = F+[1] em_0 + G_0/(lam mu-1)(EP em - EPm) - H_0/(lam mu+1)(EM em - EMm) + alpha terms
// is a linear functional of (pos_i, neg_i) with weights known top-down
// (W_0 = 1 with the mid-point form for layer 0, W_i = em_0 * prod_{1<=j<i} e_j below).
#include <type_traits>

#include "common.hpp"
#include "device_math.hpp"

namespace pz {

// ---------------------------------------------------------------------------------------------
// The layer step in three pieces, shared by the lane-per-column kernel (k_thermal_toa) and the
// cooperative kernel for small launches (k_thermal_coop).  Every operation is written out (contraction
// off, explicit fma), so both kernels -- and every angle grouping -- round identically.
// ---------------------------------------------------------------------------------------------
struct ThShared {            // angle-independent quantities of one layer
    double gam, EM, EP, q, b1, s, lam, al1, al2, al2dt, gcoef, hcoef, cmu, E;
};
struct ThAngle {             // one angle of one layer, without the running transmission W
    double e, vp, vn, c0;
};

// fluxes.py:1757-1786, 1842-1847
__device__ __forceinline__ void thermal_shared(double B0, double Bn, double dt, double w0, double g,
                                               const Exp2Coef &K, ThShared &L)
{
#pragma clang fp contract(off)
    const double mu1 = 0.5;                                    // fluxes.py:1748
    const double g1 = 2.0 - w0 * (1 + g), g2 = w0 * (1 - g);   // fluxes.py:1760
    L.lam = fsqrt(g1 * g1 - g2 * g2);                          // unfused, as numpy
    L.E = fmin(L.lam * dt, 35.0);                              // fluxes.py:1784-1786
    L.EP = fexpk(L.E, K);
    // 1/dtau, 1/g2, 1/(g1+g2) and 1/EP from one v_rcp_f64 + Newton core (prefix products, one reciprocal, two
    // multiplies per value on the way back): 17 instructions instead of 4 x 5 and three quarter-rate v_rcp_f64 fewer.
    // dtau <= ~1e4, EP <= e^35, g2 and g1+g2 of order one: the product of the four stays far inside the fp64 range.
    double idt, ig2;
    {
        const double v2 = g1 + g2;
        const double p1 = dt * g2, p2 = p1 * v2, p3 = p2 * L.EP;
        double r = frcp(p3);
        L.EM = r * p2;
        r *= L.EP;
        L.s = r * p1;                                          // fluxes.py:1766
        r *= v2;
        ig2 = r * dt;
        idt = r * g2;
    }
    L.b1 = (Bn - B0) * idt;                                    // fluxes.py:1757
    L.gam = (g1 - L.lam) * ig2;
    // fluxes.py:1772-1779 with 2 pi mu1 = pi and B0 + b1 dtau = B_{i+1}:
    //   c+up = pi B_i + q, c-up = pi B_i - q, c+dn = pi B_{i+1} + q, c-dn = pi B_{i+1} - q,
    // so the interface right-hand sides are +-(q_i - q_{i-1}) with the pi B terms cancelled
    // analytically (the reference cancels them numerically, at a cost of up to 11 digits in
    // optically thick, weakly scattering layers).
    L.q = (PI * L.b1) * L.s;
    L.cmu = (2 * PI * mu1) * fma(-L.b1, L.s, B0);
    L.al1 = (2 * PI) * fma(L.b1, L.s - mu1, B0);               // fluxes.py:1846-1847
    L.al2 = 2 * PI * L.b1;
    L.al2dt = L.al2 * dt;
    L.gcoef = (1.0 / mu1 - L.lam);                             // G = gcoef*pos   fluxes.py:1842
    L.hcoef = L.gam * (L.lam + 1.0 / mu1);                     // H = hcoef*neg   fluxes.py:1843
}

// interior layers (i > 0): fluxes.py:1877, 1897-1901
__device__ __forceinline__ void thermal_angle(const ThShared &L, double dt, double mu, double nl1,
                                              const Exp2Coef &K, ThAngle &A)
{
#pragma clang fp contract(off)
    // 1/(lam mu - 1) and 1/(lam mu + 1) from one reciprocal of the product of the two
    // 1-ulp factors (see toon_reflected.hip: no cancellation, lm1 is exact near lam mu = 1).  One Newton step (2^-46):
    // r2 is a plain factor of vp and vn -- their differences EP e - 1 and 1 - EM e are formed from full-precision values --
    // so its error stays a relative 1.4e-14 of this angle's source terms (round 5: 2 of an angle-layer's 47 instructions)
    const double lmu = L.lam * mu;
    const double lm1 = lmu - 1.0, lp1 = lmu + 1.0;
    const double r2 = frcp1(lm1 * lp1);
    const double lp = L.gcoef * (r2 * lp1), lm = L.hcoef * (r2 * lm1);
    A.e = fexp2(dt * nl1, K);
    A.vp = lp * fma(L.EP, A.e, -1.0);
    A.vn = lm * fma(-L.EM, A.e, 1.0);
    // al1 (1 - e) + al2 (mu - (dtau + mu) e) = (al1 + al2 mu)(1 - e) - (al2 dtau) e
    A.c0 = fma(fma(L.al2, mu, L.al1), 1. - A.e, -(L.al2dt * A.e));
}

// thermal_angle for NA angles at once, every statement written over the NA of them: the same operations, so the same
// bits, but the NA independent dependency chains (a 15-step polynomial and a Newton reciprocal each) alternate in the
// instruction stream instead of following one another.  hipcc keeps each angle's chain together when it schedules
// the unrolled loop over thermal_angle, and a helper wave of the cooperative kernel, which shares its SIMD with one
// other wave, then waits for its own previous instruction most of the time (toon_reflected_coop.hip has the
// measurements: ~8.7 cycles per dependent fp64 instruction against 4.2 of pipe time).
template <int NA>
__device__ __forceinline__ void thermal_angle_n(const ThShared &L, double dt, const double (&mu)[NA],
                                                const double (&nl1)[NA], const Exp2Coef &K, ThAngle (&A)[NA])
{
#pragma clang fp contract(off)
#define TH_FOR_K _Pragma("unroll") for (int k = 0; k < NA; ++k)
    double t[NA], nn[NA], f[NA], p[NA], lmu[NA], lm1[NA], lp1[NA], b[NA], y[NA], e[NA], lp[NA], lm[NA];
    TH_FOR_K t[k] = dt * nl1[k];
    TH_FOR_K lmu[k] = L.lam * mu[k];
    TH_FOR_K nn[k] = __builtin_rint(t[k]);                 // fexp2(t, K)
    TH_FOR_K lm1[k] = lmu[k] - 1.0;
    TH_FOR_K lp1[k] = lmu[k] + 1.0;
    TH_FOR_K f[k] = t[k] - nn[k];
    TH_FOR_K b[k] = lm1[k] * lp1[k];
    TH_FOR_K p[k] = fma(K.c[10], f[k], K.c[9]);
    TH_FOR_K y[k] = __builtin_amdgcn_rcp(b[k]);            // frcp1(b)
    TH_FOR_K p[k] = fma(p[k], f[k], K.c[8]);
    TH_FOR_K e[k] = fma(-b[k], y[k], 1.0);
    TH_FOR_K p[k] = fma(p[k], f[k], K.c[7]);
    TH_FOR_K y[k] = fma(y[k], e[k], y[k]);                 // r2
    TH_FOR_K p[k] = fma(p[k], f[k], K.c[6]);
    TH_FOR_K p[k] = fma(p[k], f[k], K.c[5]);
    TH_FOR_K p[k] = fma(p[k], f[k], K.c[4]);
    TH_FOR_K lp[k] = L.gcoef * (y[k] * lp1[k]);
    TH_FOR_K p[k] = fma(p[k], f[k], K.c[3]);
    TH_FOR_K lm[k] = L.hcoef * (y[k] * lm1[k]);
    TH_FOR_K p[k] = fma(p[k], f[k], K.c[2]);
    TH_FOR_K p[k] = fma(p[k], f[k], K.c[1]);
    TH_FOR_K p[k] = fma(p[k], f[k], K.c[0]);
    TH_FOR_K A[k].e = ldexp(fma(f[k], p[k], 1.0), (int)nn[k]);
    TH_FOR_K A[k].vp = lp[k] * fma(L.EP, A[k].e, -1.0);
    TH_FOR_K A[k].vn = lm[k] * fma(-L.EM, A[k].e, 1.0);
    TH_FOR_K A[k].c0 = fma(fma(L.al2, mu[k], L.al1), 1. - A[k].e, -(L.al2dt * A[k].e));
#undef TH_FOR_K
}

// top layer: mid-point form (fluxes.py:1856-1857, 1878, 1903-1907); A.e = exp(-dtau/2mu)
__device__ __forceinline__ void thermal_angle_top(const ThShared &L, double dt, double mu, double nl1,
                                                  double EPm, double EMm, const Exp2Coef &K, ThAngle &A)
{
#pragma clang fp contract(off)
    const double lmu = L.lam * mu;
    const double lm1 = lmu - 1.0, lp1 = lmu + 1.0;
    const double r2 = frcp1(lm1 * lp1);
    const double lp = L.gcoef * (r2 * lp1), lm = L.hcoef * (r2 * lm1);
    A.e = fexp2((0.5 * dt) * nl1, K);
    A.vp = lp * fma(L.EP, A.e, -EPm);
    A.vn = -lm * fma(L.EM, A.e, -EMm);
    A.c0 = fma(L.al2, fma(-(dt + mu), A.e, mu + 0.5 * dt), L.al1 * (1. - A.e));
}

struct ThSweep {             // shared sweep state: the relation everything above imposes
    double rho, delta, pgam, pEM, pq;
};

// elimination step of an interior layer: (rho, delta) of this layer, sfac and t
__device__ __forceinline__ void thermal_eliminate(ThSweep &S, double gam, double EM, double q, double &rho_n,
                                                  double &delta_n, double &sfac, double &t)
{
#pragma clang fp contract(off)
    const double em2 = S.pEM * S.pEM;
    const double a1 = fma(-(S.pgam * em2), S.rho, 1.0);
    const double a2 = fma(-em2, S.rho, S.pgam);
    const double d1 = fma(-gam, a2, a1);
    const double r12 = frcp(d1 * a1);                  // one reciprocal for 1/d1 and 1/a1
    const double inv = r12 * a1;
    const double dq = q - S.pq;
    const double rP = fma(-(S.pgam * S.pEM), S.delta, dq);
    const double rM = fma(-S.pEM, S.delta, -dq);
    rho_n = fma(gam, a1, -a2) * inv;
    delta_n = fma(a2, rP, -(a1 * rM)) * inv;
    const double ia = S.pEM * (r12 * d1);
    sfac = fma(-gam, rho_n, 1.0) * ia;
    t = fma(gam, delta_n, rP) * ia;
    (void)EM;
}

// functional update of one angle on an interior layer
__device__ __forceinline__ void thermal_accumulate(const ThAngle &A, double rho_n, double delta_n, double sfac,
                                                   double t, double &W, double &kappa, double &zeta)
{
#pragma clang fp contract(off)
    // kappa += zeta t + W (c0 + vn delta_n);  zeta = zeta sfac + W (vp - vn rho_n): the layer's terms first, one
    // multiply by the running transmission each (7 instructions; 9 with vp, vn, c0 scaled by W one by one)
    const double u = fma(A.vn, delta_n, A.c0), v = fma(-A.vn, rho_n, A.vp);
    kappa = fma(W, u, fma(zeta, t, kappa));
    zeta = fma(W, v, zeta * sfac);
    W = W * A.e;
}

// boundary intensity F+[n] (fluxes.py:1871-1873, 2256, 2310-2312)
template <bool IS3D>
__device__ __forceinline__ double thermal_bottom(double Bn, double b1, double mu, double rs, int hard_surface)
{
#pragma clang fp contract(off)
    if (!IS3D) {
        if (hard_surface) return ((1.0 - rs) * Bn) * (2 * PI);       // fluxes.py:1871
        return fma(b1, mu, Bn) * (2 * PI);                           // fluxes.py:1873
    }
    if (hard_surface) return PI * (PI * Bn);                         // fluxes.py:2256,2310
    return PI * fma(b1, mu, Bn);                                     // fluxes.py:2312
}

// surface row b_surface - c+dn + rs c-dn with the pi B_n terms cancelled analytically:
//   1-D soft (fluxes.py:1806) : pi [b1 (mu1 - s) + rs (B_n - b1 s)]
//   1-D hard (fluxes.py:1803) : -pi b1 s (1 + rs)                (b_surface = (1-rs) pi B_n)
//   3-D soft (fluxes.py:2258) : same as 1-D soft
//   3-D hard (fluxes.py:2256) : pi [rs B_n - b1 s (1 + rs)]     (b_surface = pi B_n, no emissivity)
template <bool IS3D>
__device__ __forceinline__ double thermal_surface_pos(const ThSweep &S, double Bn, double b1_last, double s_last,
                                                      double rs, int hard_surface)
{
#pragma clang fp contract(off)
    const double mu1 = 0.5;
    double bsum;
    const double bs_l = b1_last * s_last;
    if (!hard_surface) bsum = PI * fma(rs, Bn - bs_l, b1_last * (mu1 - s_last));
    else if (!IS3D) bsum = -PI * (bs_l * (1.0 + rs));
    else bsum = PI * fma(rs, Bn, -(bs_l * (1.0 + rs)));
    const double em2 = S.pEM * S.pEM;
    const double egr = em2 * (S.pgam - rs);
    return fma(S.pEM, bsum, -(egr * S.delta)) / fma(-egr, S.rho, fma(-rs, S.pgam, 1.0));
}

// `u1p`: the angle table (the kernel argument's or the batch entry's); `wgt`, `wgt2`: the disk weights, always the
// kernel argument's (indexed by the angle group, so they must not travel in a patched copy of the arguments)
template <int NA, bool IS3D, typename U1Ptr>
__device__ __forceinline__ void thermal_toa_body(const ThermalArgs &a, U1Ptr u1p, const double *wgt, const double *wgt2)
{
#pragma clang fp contract(off)
    const long col = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (col >= a.ncol) return;
    const int nfac = IS3D ? a.nfac : 1;
    const long w = IS3D ? col / nfac : (a.ncolper > 1 ? col / a.ncolper : col);
    const int fac = IS3D ? (int)(col - w * nfac) : 0;
    const int n = a.nlayer;
    const long pitch = a.pitch;
    const double mu1 = 0.5;                                   // fluxes.py:1748
    const double wn = a.wno[w];
    const double dwn = (!IS3D && a.calc_type == 1) ? a.dwno[w] : 0.0;
    const double rs = a.surf_reflect[w];
    const bool integrated = (!IS3D && a.calc_type == 1);
    const long lstride = IS3D ? nfac : 1;                     // tlevel_3d is (nlevel, ng, nt)
    const double *tl = a.tlevel + fac, *pl = a.plevel + fac;

    Exp2Coef K;
    K.load();
    double u1[NA], nl1[NA];                                    // nl1 = -log2(e)/u1: exp(-x/u1) = 2^(x nl1)
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        u1[k] = IS3D ? a.u1_tab[fac] : u1p[blockIdx.y * NA + k];    // 3-D thermal takes ubar1 as is
        nl1[k] = NEG_LOG2E / u1[k];
    }
    const double *p_dtau = a.dtau + col, *p_w0 = a.w0 + col, *p_cosb = a.cosb + col;

    auto planck = [&](int l) {
        const double t = tl[(long)l * lstride];
        return integrated ? planck_integrated(t, wn, dwn) : planck_lambda(t, wn, K);
    };

    double W[NA], kappa[NA], zeta[NA];
    ThSweep S{0.0, 0.0, 0.0, 0.0, 0.0};
    double b1_last = 0.0, s_last = 0.0;
    double Bn = planck(0);
    const double B_top = Bn;

    // 3-D entry point: cosb NULL = column without cloud (cosb_og = 0 everywhere, optics.py:338): not read
    const bool has_g = !IS3D || a.cosb != nullptr;
    double n_dt = p_dtau[0], n_w0 = p_w0[0], n_cb = has_g ? p_cosb[0] : 0.0;
    // One layer.  FIRST (layer 0, the mid-point form) is peeled off at compile time.  (Two interior layers per trip, so that
    // the 7 v_mov_b64 of the sweep state at the loop's back edge go: measured SLOWER, 0.1573 against 0.1555 ms at 1e5
    // columns -- the doubled body is 27 KB of code.)  Same operations: same bits.
    auto layer = [&](const int i, auto first_c) {
        constexpr bool FIRST = decltype(first_c)::value;
        const double dt = n_dt, w0 = n_w0, g = n_cb;
        if (i + 1 < n) {
            const long o = (long)(i + 1) * pitch;
            n_dt = p_dtau[o];
            n_w0 = p_w0[o];
            if (has_g) n_cb = p_cosb[o];
        }
        const double B0 = Bn;
        Bn = planck(i + 1);
        ThShared L;
        thermal_shared(B0, Bn, dt, w0, g, K, L);
        double rho_n = L.gam, delta_n = 0.0, sfac = 0.0, t = 0.0;
        double EPm = 0.0, EMm = 0.0;
        if constexpr (FIRST) {
            const double tau_top = dt * pl[0] / (pl[lstride] - pl[0]);      // fluxes.py:1797
            const double b_top = IS3D ? PI * (1.0 - fexpk(-tau_top / mu1, K)) * B_top   // :2253
                                      : (1.0 - fexpk(-tau_top / mu1, K)) * B_top * PI;  // :1800
            delta_n = b_top - L.cmu;
            EPm = fexpk(0.5 * L.E, K);                         // fluxes.py:1856-1857
            EMm = frcp(EPm);
        } else {
            thermal_eliminate(S, L.gam, L.EM, L.q, rho_n, delta_n, sfac, t);
        }
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            ThAngle A;
            if constexpr (FIRST) {
                thermal_angle_top(L, dt, u1[k], nl1[k], EPm, EMm, K, A);
                kappa[k] = fma(A.vn, delta_n, A.c0);
                zeta[k] = fma(-A.vn, rho_n, A.vp);
                W[k] = A.e;
            } else {
                thermal_angle(L, dt, u1[k], nl1[k], K, A);
                thermal_accumulate(A, rho_n, delta_n, sfac, t, W[k], kappa[k], zeta[k]);
            }
        }
        S.rho = rho_n;
        S.delta = delta_n;
        S.pgam = L.gam;
        S.pEM = L.EM;
        S.pq = L.q;
        b1_last = L.b1;
        s_last = L.s;
    };
    layer(0, std::true_type{});
    for (int i = 1; i < n; ++i) layer(i, std::false_type{});
    // The boundary intensity F+[n] enters through the transmission down to the bottom (for a single layer it feeds the
    // mid-point of layer 0 directly): the last layer's statement, AFTER the loop -- inside it the compiler turned
    // `if (i == n - 1)` into selects and evaluated both forms of thermal_bottom for every angle of every layer (4
    // v_cndmask + 4 fp64 per angle-layer: 40 of the 421 instructions of a five-angle layer, profiles/r05_isa_*.json).
    // Same operands, same operation: same bits.
#pragma unroll
    for (int k = 0; k < NA; ++k)
        kappa[k] = fma(W[k], thermal_bottom<IS3D>(Bn, b1_last, u1[k], rs, a.hard_surface), kappa[k]);
    const double pos = thermal_surface_pos<IS3D>(S, Bn, b1_last, s_last, rs, a.hard_surface);
    // one running disk sum over all angles (disco.py:174-176): a later angle chunk continues from the stored one
    double disk = (!IS3D && a.disk && !a.disk_first) ? a.disk[w] : 0.0;
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const double x = fma(zeta[k], pos, kappa[k]);
        if (IS3D) a.flux[(long)fac * a.nwno + w] = x;
        else a.flux[(long)(blockIdx.y * NA + k) * a.ncol + col] = x;
        {   // flux + x*gweight*tweight in the reference's order, unfused (disco.py:174-176), as k_compress
            const int ia = IS3D ? 0 : blockIdx.y * NA + k;
            disk = disk + x * wgt[ia] * wgt2[ia];
        }
    }
    if (!IS3D && a.disk) {                                     // fused disco.compress_thermal
        double acc = disk;
        if (a.disk_last) acc = acc * a.disk_scale;
        a.disk[w] = acc;
    }
}

template <int NA, bool IS3D>
__global__ __launch_bounds__(256) void k_thermal_toa(const ThermalArgs a)
{
    thermal_toa_body<NA, IS3D>(a, a.u1, a.wgt, a.wgt2);
}

// `nspec` spectra of one shape in one grid (picaso_get_thermal_1d_batch_dev): grid.z = spectrum, whose planes,
// level tables, outputs and angles come from the device table a.batch (read through the constant address
// space like kernel arguments, see k_reflected_toa_batch).  Same body, same bits.
template <int NA, bool IS3D>
__global__ __launch_bounds__(256) void k_thermal_toa_batch(const ThermalArgs a)
{
    typedef const __attribute__((address_space(4))) ThermalBatchItem *ItemPtr;
    const auto &it = *((ItemPtr)(unsigned long)a.batch + blockIdx.z);
    ThermalArgs b = a;
    b.dtau = it.dtau; b.w0 = it.w0; b.cosb = it.cosb; b.surf_reflect = it.surf_reflect;
    b.tlevel = it.tlevel; b.plevel = it.plevel; b.flux = it.flux; b.disk = it.disk;
    b.u1_tab = it.u1_tab;
    thermal_toa_body<NA, IS3D>(b, it.u1, a.wgt, a.wgt2);
}

// ---------------------------------------------------------------------------------------------
// Cooperative kernel for small launches (BASELINE configs[1]: 1e4 columns; wavelength shards of a
// multi-GPU run; climate-sized grids).  With one lane per column a 1e4-column launch is 157 waves (785
// with every angle as its own wave) on 1 024 SIMDs: the chip is nearly empty and the launch lasts as
// long as ONE wave needs for its 90-layer sweep, most of which is not sequential at all -- the Planck
// function, lambda, Gamma, the exponentials and the per-angle source terms of a layer do not depend on
// the layers above.  Here a workgroup of eight waves owns 64 columns: six HELPER waves compute those
// layer quantities, one layer each per round, into an LDS buffer ([variable][lane]: conflict-free
// 8-byte accesses), a PLANCK wave evaluates the Planck function at the levels two rounds ahead (so
// a helper does not evaluate it twice per layer), and the SWEEPER wave runs the short sequential part
// (the two-stream elimination, ~20 instructions, and 9 per angle) over the layers of the previous
// round; two layer buffers, one workgroup barrier per round of six layers.  Same three layer pieces as k_thermal_toa: bit-identical
// results.
// ---------------------------------------------------------------------------------------------
// helper waves: the NA angles of a layer statement by statement (thermal_angle_n) instead of angle by angle
#ifndef PZ_COOP_ANGLES_INTERLEAVED
#define PZ_COOP_ANGLES_INTERLEAVED 1
#endif
#ifndef PZ_COOP_HELPERS
#define PZ_COOP_HELPERS 6
#endif
#ifndef PZ_COOP_SWEEPER
#define PZ_COOP_SWEEPER 3                          // hardware wave index of the sweeper (see k_thermal_coop)
#endif
constexpr int COOP_HELPERS = PZ_COOP_HELPERS;     // helper waves = layers per round
constexpr int COOP_MAX_LEVELS = 256;              // level temperatures carried in the kernel arguments
constexpr int COOP_SHARED = 3;                    // gam, EM, q (b1 and s of the last layer travel separately)

// the level temperatures and the two top pressures travel as kernel arguments: no table upload (a
// host-to-device copy on the stream ahead of every launch) for a launch that lasts 30 microseconds
struct ThermalCoopArgs {
    ThermalArgs base;
    double tlevel[COOP_MAX_LEVELS];
    double p0, p1;
};

template <int NA>
__global__ __launch_bounds__(64 * (COOP_HELPERS + 2)) void k_thermal_coop(const ThermalCoopArgs ca)
{
#pragma clang fp contract(off)
    const ThermalArgs &a = ca.base;
    constexpr int NV = COOP_SHARED + 4 * NA;      // doubles per (layer, column)
    __shared__ double buf[2][COOP_HELPERS][NV][64];
    __shared__ double last_bs[2][64];             // b1, s of the bottom layer (surface row)
    __shared__ double Bbuf[3][COOP_HELPERS + 1][64];   // Planck function at the levels of a round, two rounds ahead
    const int lane = threadIdx.x & 63;
    // role: the waves of a workgroup are placed on the CU's four SIMDs round-robin, so with seven waves
    // wave 3 has its SIMD to itself: that one sweeps (a lone fp64 wave issues faster than one of two)
    const int hw_wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // 0 = sweeper, 1 .. H = helpers, H + 1 = the Planck wave
    const int wave = (hw_wave == PZ_COOP_SWEEPER) ? 0 : (hw_wave < PZ_COOP_SWEEPER ? hw_wave + 1 : hw_wave);
    long col = (long)blockIdx.x * 64 + lane;
    const bool active = col < a.ncol;
    if (!active) col = a.ncol - 1;                // padding lanes shadow the last column, never store
    const long w = a.ncolper > 1 ? col / a.ncolper : col;
    const int n = a.nlayer;
    const long pitch = a.pitch;
    const double mu1 = 0.5;
    const double wn = a.wno[w];
    const bool integrated = (a.calc_type == 1);
    const double dwn = integrated ? a.dwno[w] : 0.0;
    const double rs = a.surf_reflect[w];
    const double *tl = ca.tlevel;
    const double pl[2] = {ca.p0, ca.p1};
    Exp2Coef K;
    K.load();
    double u1[NA], nl1[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        u1[k] = a.u1[k];
        nl1[k] = NEG_LOG2E / u1[k];
    }
    const double *p_dtau = a.dtau + col, *p_w0 = a.w0 + col, *p_cosb = a.cosb + col;
    auto planck = [&](int l) {
        const double t = tl[l];
        return integrated ? planck_integrated(t, wn, dwn) : planck_lambda(t, wn, K);
    };
    const int nrounds = (n - 1 + COOP_HELPERS - 1) / COOP_HELPERS;     // layers 1 .. n-1 in rounds of five

    // Planck wave: the H + 1 levels of round r (layer i reads B_i and B_{i+1}) into Bbuf[r % 3]
    auto fill_planck = [&](int r) {
#pragma unroll
        for (int j = 0; j <= COOP_HELPERS; ++j) {
            const int l = 1 + r * COOP_HELPERS + j;
            if (l <= n) Bbuf[r % 3][j][lane] = planck(l);
        }
    };
    // helpers: layer 1 + r*H + (wave-1) of round r into buf[r & 1][wave-1]; the three plane values of
    // the helper's layer of the NEXT round are loaded before this round's arithmetic (a helper starts
    // its round right after a barrier: an unhidden HBM latency per round otherwise)
    double pf_dt = 0.0, pf_w0 = 0.0, pf_g = 0.0;
    auto prefetch = [&](int r) {
        const int i = 1 + r * COOP_HELPERS + (wave - 1);
        if (i >= n) return;
        const long o = (long)i * pitch;
        pf_dt = p_dtau[o];
        pf_w0 = p_w0[o];
        pf_g = p_cosb[o];
    };
    auto produce = [&](int r) {
        const int i = 1 + r * COOP_HELPERS + (wave - 1);
        const double dt = pf_dt, w0 = pf_w0, g = pf_g;
        prefetch(r + 1);
        if (i >= n) return;
        ThShared L;
        thermal_shared(Bbuf[r % 3][wave - 1][lane], Bbuf[r % 3][wave][lane], dt, w0, g, K, L);
        double(*slot)[64] = buf[r & 1][wave - 1];
        slot[0][lane] = L.gam;
        slot[1][lane] = L.EM;
        slot[2][lane] = L.q;
        if (i == n - 1) {
            last_bs[0][lane] = L.b1;
            last_bs[1][lane] = L.s;
        }
        ThAngle A[NA];
#if PZ_COOP_ANGLES_INTERLEAVED
        thermal_angle_n<NA>(L, dt, u1, nl1, K, A);
#else
#pragma unroll
        for (int k = 0; k < NA; ++k) thermal_angle(L, dt, u1[k], nl1[k], K, A[k]);
#endif
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            slot[COOP_SHARED + 4 * k + 0][lane] = A[k].e;
            slot[COOP_SHARED + 4 * k + 1][lane] = A[k].vp;
            slot[COOP_SHARED + 4 * k + 2][lane] = A[k].vn;
            slot[COOP_SHARED + 4 * k + 3][lane] = A[k].c0;
        }
    };

    double W[NA], kappa[NA], zeta[NA];
    ThSweep S{0.0, 0.0, 0.0, 0.0, 0.0};
    double b1_last = 0.0, s_last = 0.0;
    if (wave == 0) {
        // the sweeper does the top layer (mid-point form) itself while the helpers fill round 0
        const double dt = p_dtau[0], w0 = p_w0[0], g = p_cosb[0];
        const double B_top = planck(0);
        ThShared L;
        thermal_shared(B_top, planck(1), dt, w0, g, K, L);
        const double tau_top = dt * pl[0] / (pl[1] - pl[0]);            // fluxes.py:1797
        const double b_top = (1.0 - fexpk(-tau_top / mu1, K)) * B_top * PI; // :1800
        const double delta_n = b_top - L.cmu;
        const double EPm = fexpk(0.5 * L.E, K);                         // fluxes.py:1856-1857
        const double EMm = frcp(EPm);
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            ThAngle A;
            thermal_angle_top(L, dt, u1[k], nl1[k], EPm, EMm, K, A);
            kappa[k] = fma(A.vn, delta_n, A.c0);
            zeta[k] = fma(-A.vn, L.gam, A.vp);
            W[k] = A.e;
        }
        S.rho = L.gam;
        S.delta = delta_n;
        S.pgam = L.gam;
        S.pEM = L.EM;
        S.pq = L.q;
        b1_last = L.b1;
        s_last = L.s;
    } else if (wave == COOP_HELPERS + 1) {
        fill_planck(0);
    } else {
        prefetch(0);
    }
    __syncthreads();
    if (wave >= 1 && wave <= COOP_HELPERS) produce(0);
    else if (wave == COOP_HELPERS + 1 && nrounds > 1) fill_planck(1);
    __syncthreads();
    for (int r = 0; r < nrounds; ++r) {
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < COOP_HELPERS; ++j) {
                const int i = 1 + r * COOP_HELPERS + j;
                if (i < n) {
                    double(*slot)[64] = buf[r & 1][j];
                    const double gam = slot[0][lane], EM = slot[1][lane], q = slot[2][lane];
                    double rho_n, delta_n, sfac, t;
                    thermal_eliminate(S, gam, EM, q, rho_n, delta_n, sfac, t);
#pragma unroll
                    for (int k = 0; k < NA; ++k) {
                        ThAngle A;
                        A.e = slot[COOP_SHARED + 4 * k + 0][lane];
                        A.vp = slot[COOP_SHARED + 4 * k + 1][lane];
                        A.vn = slot[COOP_SHARED + 4 * k + 2][lane];
                        A.c0 = slot[COOP_SHARED + 4 * k + 3][lane];
                        thermal_accumulate(A, rho_n, delta_n, sfac, t, W[k], kappa[k], zeta[k]);
                    }
                    S.rho = rho_n;
                    S.delta = delta_n;
                    S.pgam = gam;
                    S.pEM = EM;
                    S.pq = q;
                }
            }
        } else if (wave <= COOP_HELPERS) {
            if (r + 1 < nrounds) produce(r + 1);
        } else if (r + 2 < nrounds) {
            fill_planck(r + 2);
        }
        __syncthreads();
    }
    if (wave != 0) return;
    if (n > 1) {
        b1_last = last_bs[0][lane];
        s_last = last_bs[1][lane];
    }
    const double Bn = planck(n);
#pragma unroll
    for (int k = 0; k < NA; ++k)                               // F+[n] boundary intensity
        kappa[k] = fma(W[k], thermal_bottom<false>(Bn, b1_last, u1[k], rs, a.hard_surface), kappa[k]);
    const double pos = thermal_surface_pos<false>(S, Bn, b1_last, s_last, rs, a.hard_surface);
    double disk = (a.disk && active && !a.disk_first) ? a.disk[w] : 0.0;
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const double x = fma(zeta[k], pos, kappa[k]);
        if (active) a.flux[(long)k * a.ncol + col] = x;
        disk = disk + x * a.wgt[k] * a.wgt2[k];                // reference order, unfused (disco.py:174-176)
    }
    if (a.disk && active) {                                    // fused disco.compress_thermal
        double acc = disk;
        if (a.disk_last) acc = acc * a.disk_scale;
        a.disk[w] = acc;
    }
}

template <int NA>
static int launch_coop(picaso_ctx *ctx, const ThermalCoopArgs &a)
{
    const dim3 grid((unsigned)((a.base.ncol + 63) / 64));
    hipLaunchKernelGGL((k_thermal_coop<NA>), grid, dim3(64 * (COOP_HELPERS + 2)), 0, ctx->stream, a);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

// Small launch: cooperative kernel (all angles in one sweeper, fused disk sum)?  The caller decides on
// the column count; here only the shape constraints.
bool thermal_coop_ok(const ThermalArgs &a)
{
    return a.na >= 1 && a.na <= 5 && a.ny <= 1 && a.nlayer >= 2 && a.nlayer + 1 <= COOP_MAX_LEVELS &&
           !getenv("PICASO_AMD_THERMAL_NO_COOP");
}

// tlevel / plevel: HOST arrays (nlevel) here
int launch_thermal_coop(picaso_ctx *ctx, const ThermalArgs &a, const double *tlevel_host, const double *plevel_host)
{
    ThermalCoopArgs ca{};
    ca.base = a;
    for (int l = 0; l <= a.nlayer; ++l) ca.tlevel[l] = tlevel_host[l];
    ca.p0 = plevel_host[0];
    ca.p1 = plevel_host[1];
    switch (a.na) {
        case 1: return launch_coop<1>(ctx, ca);
        case 2: return launch_coop<2>(ctx, ca);
        case 3: return launch_coop<3>(ctx, ca);
        case 4: return launch_coop<4>(ctx, ca);
        case 5: return launch_coop<5>(ctx, ca);
    }
    return fail(ctx, "thermal: unsupported angle count %d for the cooperative kernel", a.na);
}

template <int NA>
static int launch1d(picaso_ctx *ctx, const ThermalArgs &a)
{
    const int block = 256;
    const dim3 grid((unsigned)((a.ncol + block - 1) / block), (unsigned)(a.ny > 1 ? a.ny : 1),
                    a.batch ? (unsigned)a.nspec : 1u);
    if (a.batch) hipLaunchKernelGGL((k_thermal_toa_batch<NA, false>), grid, dim3(block), 0, ctx->stream, a);
    else hipLaunchKernelGGL((k_thermal_toa<NA, false>), grid, dim3(block), 0, ctx->stream, a);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

int launch_thermal_toa(picaso_ctx *ctx, const ThermalArgs &a, bool is3d)
{
    if (a.ncol <= 0 || a.nlayer < 1) return fail(ctx, "thermal: empty problem");
    if (is3d) {
        const int block = 256;
        const long grid = (a.ncol + block - 1) / block;
        if (a.batch)
            hipLaunchKernelGGL((k_thermal_toa_batch<1, true>), dim3((unsigned)grid, 1u, (unsigned)a.nspec), dim3(block),
                               0, ctx->stream, a);
        else
            hipLaunchKernelGGL((k_thermal_toa<1, true>), dim3((unsigned)grid), dim3(block), 0,
                           ctx->stream, a);
        PZ_HIP(ctx, hipGetLastError());
        return 0;
    }
    switch (a.na) {
        case 1: return launch1d<1>(ctx, a);
        case 2: return launch1d<2>(ctx, a);
        case 3: return launch1d<3>(ctx, a);
        case 4: return launch1d<4>(ctx, a);
        case 5: return launch1d<5>(ctx, a);
        case 6: return launch1d<6>(ctx, a);
        case 7: return launch1d<7>(ctx, a);
        case 8: return launch1d<8>(ctx, a);
    }
    return fail(ctx, "thermal: unsupported angle chunk %d", a.na);
}

}  // namespace pz
