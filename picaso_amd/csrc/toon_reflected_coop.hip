// Toon89 two-stream reflected light for SMALL launches: one workgroup per 64 wavelength columns -- gfx950.
//
// Replaces fluxes.get_reflected_1d (reference picaso/fluxes.py:1009-1413, get_toa_intensity=1, get_lvl_flux=0,
// the reference's default options) where a launch has at most one 64-column block per CU (<= 16 384 columns on
// the MI355X): a wavelength shard of a multi-GPU run (12 500 columns per GPU on eight), a climate grid, a coarse
// spectrum.
//
// k_reflected_toa (toon_reflected.hip) carries all disk angles of a column in one lane: the right shape when
// every SIMD has two such waves, but a launch of <= 1 workgroup per CU lasts as long as ONE lane's serial
// instruction stream (~480 fp64 instructions per layer at five angles), and spreading the angles over separate
// workgroups (api.hip:reflected_angle_group) repeats the angle-independent part of the work in every one of
// them.  Here the waves of ONE workgroup share a column block instead:
//   * wave L ("loader") streams the eleven plane values of every layer from HBM into an LDS ring, two rounds of
//     RC_R layers ahead of their use (a round's loads stay in flight across a whole step), and evaluates the
//     wave-uniform shortcut tests of k_reflected_toa's `prep` on the values it holds;
//   * wave S ("shared") walks down the layers a round behind L and computes everything that does not depend on
//     the disk angle -- the two-stream gammas, lambda, Gamma, exp(+-lambda dtau), the single-scattering phase
//     function, the elimination factors (the only sequential part of it: the recurrence rho) -- into a second ring
//     ([layer][variable][lane]: conflict-free 8-byte accesses);
//   * waves A_k, one per disk angle, run the per-angle part of the sweep a round behind S (the direct-beam terms,
//     the source-function integrals, the TOA functional: `reflected_layer`'s angle loop body, one iteration each)
//     with their seven state variables in registers;
//   * one workgroup barrier per round; the disk sum (disco.compress_disco) is fused: the angle waves leave their
//     result in LDS and wave S adds them in the reference's (g, t) order.
// Every floating-point operation is the one reflected_layer<.., FAST = true> performs for that (column, angle), in
// the same order (contraction off, explicit fma): the results are BIT-IDENTICAL to the fused five-angle launch,
// so a spectrum does not depend on how many GPUs it was cut across (tests/test_parity_gpu.py,
// test_fullsize_gpu.py, test_fuzz_gpu.py all run through this kernel at their sizes).
//
// What it buys, measured (the measurements behind it: DESIGN.md appendix A.3): 0.047-0.048 ms for any launch up to 16 384 columns x 90 layers x 5
// angles, against 0.048-0.050 (one angle per workgroup, <= 13 056 columns) and 0.061 (angle pairs, to 16 384) of the
// grid.y shapes.  Why not more: a wave alone on its SIMD issues one fp64 instruction per ~7.5 cycles here (2.38 GHz
// in these light launches), two waves sharing a SIMD one per ~5.8 between them, against the pipe's 4.2 -- the
// kernel's 556 fp64 instructions per column-block-layer on 7 waves last ~1 200 cycles where 4 saturated SIMDs would
// need 584.  Getting there takes >= 3 balanced waves per SIMD (PZ_RCOOP_APW = 2, two angles interleaved statement by
// statement in one wave, measured slower: 0.051 ms).
#include <type_traits>

#include "common.hpp"
#include "device_math.hpp"

namespace pz {

#ifndef PZ_RCOOP_ROUND
#define PZ_RCOOP_ROUND 4
#endif
constexpr int RC_R = PZ_RCOOP_ROUND;          // layers per round
// Hardware wave index of wave S and of the loader wave L: the waves of a workgroup go to the CU's four SIMDs
// round-robin, so with seven waves (five angles) SIMD 2 holds waves 2 and 6, SIMD 3 only wave 3.  S is the
// longest role and shares its SIMD with L, which only moves data.
#ifndef PZ_RCOOP_SWAVE
#define PZ_RCOOP_SWAVE 2
#endif
// the eleven plane values of a layer, as wave L leaves them in LDS
enum { RW_DT = 0, RW_TAUN, RW_W0, RW_G, RW_GCOS2, RW_FC, RW_FR, RW_DTO, RW_TAUO, RW_W0O, RW_CBO, RW_NV };
// what wave S leaves for the angle waves
enum {
    RC_LAM = 0, RC_EP, RC_EM, RC_GAM, RC_FW0H, RC_A0, RC_W2PI, RC_SSAH,
    RC_A1I, RC_A2I, RC_IA, RC_SFAC, RC_RHON,         // read for every layer
    RC_A1, RC_C15, RC_GMC,                           // layers with cloud in some lane
    RC_NV
};
enum { RCF_CUM_TAU = 1, RCF_EO_OK = 2, RCF_SAME_DT = 4, RCF_NOCLD = 8, RCF_ALL = 15 };
#ifndef PZ_RCOOP_APW
#define PZ_RCOOP_APW 1
#endif
constexpr int RC_APW = PZ_RCOOP_APW;                 // angles per angle wave (1 or 2; measured: 1)
constexpr int RC_MAX_ANGLES = 6 * RC_APW;            // at most 6 angle waves + S + L = 8 waves (two per SIMD)

struct RcState {
    double T, XU, EO, KAPPA, ZETA, D1, D2;      // the seven sweep variables of reflected_layer, for one angle
    double l_gam, l_EM, l_rho;                  // of the last layer, for the surface row
};

// NA-wide forms of the math helpers: the same operations as fexp2 / frcp (device_math.hpp) for NA independent
// arguments, written statement by statement over the NA of them.  hipcc keeps long dependent chains together when it
// schedules a basic block; a wave that is alone on its SIMD then issues one fp64 instruction per ~8.7 cycles (each
// waits for its predecessor) where the pipe takes one per ~4.2.  Interleaved in the source, the chains of the NA
// angles fill each other's latency.
#define RC_FOR_K _Pragma("unroll") for (int kk = 0; kk < NA; ++kk)
template <int NA>
__device__ __forceinline__ void fexp2_n(const double (&t)[NA], const Exp2Coef &K, double (&out)[NA])
{
#pragma clang fp contract(off)
    double nn[NA], f[NA], p[NA];
    RC_FOR_K nn[kk] = __builtin_rint(t[kk]);
    RC_FOR_K f[kk] = t[kk] - nn[kk];
    RC_FOR_K p[kk] = fma(K.c[10], f[kk], K.c[9]);
#pragma unroll
    for (int i = 8; i >= 0; --i) RC_FOR_K p[kk] = fma(p[kk], f[kk], K.c[i]);
    RC_FOR_K out[kk] = ldexp(fma(f[kk], p[kk], 1.0), (int)nn[kk]);
}
template <int NA>
__device__ __forceinline__ void frcp_n(const double (&b)[NA], double (&y)[NA])
{
#pragma clang fp contract(off)
    double e[NA];
    RC_FOR_K y[kk] = __builtin_amdgcn_rcp(b[kk]);
    RC_FOR_K e[kk] = fma(-b[kk], y[kk], 1.0);
    RC_FOR_K y[kk] = fma(y[kk], e[kk], y[kk]);
    RC_FOR_K e[kk] = fma(-b[kk], y[kk], 1.0);
    RC_FOR_K y[kk] = fma(y[kk], e[kk], y[kk]);
}
// frcp1 (one Newton step, 2^-46) over NA arguments: the per-angle 1/(den (lu - 1)(lu + 1)) of PZ_REFL_DIET
template <int NA>
__device__ __forceinline__ void frcp1_n(const double (&b)[NA], double (&y)[NA])
{
#pragma clang fp contract(off)
    double e[NA], y0[NA];
    RC_FOR_K y0[kk] = __builtin_amdgcn_rcp(b[kk]);
    RC_FOR_K e[kk] = fma(-b[kk], y0[kk], 1.0);
    RC_FOR_K y[kk] = fma(y0[kk], e[kk], y0[kk]);
}

// One layer of NA angles: NA iterations of reflected_layer's angle loop, every statement written over the NA angles
// (see above).  `in`: the layer's plane values (wave L), `slot`: the angle-independent quantities (wave S), read
// from LDS once for all NA; `fl`: the wave-uniform shortcut flags of the layer.  PLAIN: all four flags are set,
// known at compile time (no cloud terms, no fall-back exponentials, no branches).
template <bool ZP, bool FIRST, bool LAST, bool PLAIN, int NA>
__device__ __forceinline__ void rc_angle(const double (*in)[64], const double (*slot)[64], int fl, int lane,
                                         const ReflectedArgs::Angle (&g)[NA], const Exp2Coef &K, double b_top,
                                         RcState (&st)[NA])
{
#pragma clang fp contract(off)
    constexpr bool first = FIRST, last = LAST;
    const bool cum_tau = PLAIN || (fl & RCF_CUM_TAU), eo_ok = PLAIN || (fl & RCF_EO_OK),
               same_dt = PLAIN || (fl & RCF_SAME_DT), nocld = PLAIN || (fl & RCF_NOCLD);
    const double lam = slot[RC_LAM][lane], EP = slot[RC_EP][lane], EM = slot[RC_EM][lane];
    const double gam = slot[RC_GAM][lane], dt = in[RW_DT][lane], Fw0h = slot[RC_FW0H][lane];
    const double A0 = slot[RC_A0][lane], w2pi = slot[RC_W2PI][lane], ssa_h = slot[RC_SSAH][lane];
    const double gcq = in[RW_GCOS2][lane], a1i = slot[RC_A1I][lane], a2i = slot[RC_A2I][lane];
    const double ia = slot[RC_IA][lane], sfac = slot[RC_SFAC][lane], rho_n = slot[RC_RHON][lane];
    double A1 = 0.0, c15 = 0.0, gmc = 0.0;
    if (!nocld) {
        A1 = slot[RC_A1][lane];
        c15 = slot[RC_C15][lane];
        gmc = slot[RC_GMC][lane];
    }
    double tau_n = 0.0, tauo = 0.0, dto = 0.0;
    if (!PLAIN) {
        if (!cum_tau) tau_n = in[RW_TAUN][lane];
        if (first || !eo_ok) tauo = in[RW_TAUO][lane];
        if (!same_dt) dto = in[RW_DTO][lane];
    }
    const double lam2 = lam * lam;                 // toon_gammas forms it the same way
    const double gp = 1.0 + gam;
    const double gEM = gam * EM;
    // ---- the two long chains of a layer: exp(-dtau/u1) and the one reciprocal for 1/den, 1/(lu-1), 1/(lu+1) ----
    double targ[NA], et[NA], e0[NA], den[NA], lu[NA], lm1[NA], lp1[NA], lml[NA], q3[NA], r3[NA];
    RC_FOR_K targ[kk] = dt * g[kk].nl1;
    RC_FOR_K den[kk] = sub_unfused(lam2, g[kk].iu0sq);
    RC_FOR_K lu[kk] = lam * g[kk].u1;
    RC_FOR_K lm1[kk] = lu[kk] - 1.0;
    RC_FOR_K lp1[kk] = lu[kk] + 1.0;
    RC_FOR_K lml[kk] = lm1[kk] * lp1[kk];
    RC_FOR_K q3[kk] = den[kk] * lml[kk];
    fexp2_n<NA>(targ, K, et);
    if (PZ_REFL_DIET) frcp1_n<NA>(q3, r3);
    else frcp_n<NA>(q3, r3);
    if (ZP) {
        RC_FOR_K e0[kk] = et[kk];
    } else {
        double t0[NA];
        RC_FOR_K t0[kk] = dt * g[kk].nl0;
        fexp2_n<NA>(t0, K, e0);
    }
    double rden[NA], rd[NA], hz[NA], am2[NA], ap2[NA], xd[NA], fw[NA], fx[NA], fxd[NA];
    RC_FOR_K rden[kk] = r3[kk] * lml[kk];
    RC_FOR_K rd[kk] = r3[kk] * den[kk];
    // PLAIN: the cloud terms folded by hand (fma(0, y, z) = z, x + 0 = x: exact), as reflected_layer's NC copy
    RC_FOR_K hz[kk] = PLAIN ? g[kk].iu0 : fma(A1, ZP ? g[kk].u1 : g[kk].u0, g[kk].iu0);
    RC_FOR_K am2[kk] = A0 + hz[kk];
    RC_FOR_K ap2[kk] = A0 - hz[kk];
    if (cum_tau) {
        RC_FOR_K xd[kk] = st[kk].XU * e0[kk];
    } else {
        RC_FOR_K xd[kk] = fexp2_cold(tau_n * (ZP ? g[kk].nl1 : g[kk].nl0), K);
    }
    RC_FOR_K fw[kk] = Fw0h * rden[kk];
    RC_FOR_K fx[kk] = fw[kk] * st[kk].XU;
    double cmu[NA], cpu[NA], cmd[NA], cpd[NA], B0[NA], Aqq[NA], X[NA], Y[NA], Tw[NA], Trd[NA], ee[NA], ff[NA];
    RC_FOR_K cmu[kk] = am2[kk] * fx[kk];
    RC_FOR_K cpu[kk] = ap2[kk] * fx[kk];
    if (PZ_REFL_DIET && cum_tau) {                      // reflected_layer, PZ_REFL_DIET (1)
        RC_FOR_K cmd[kk] = cmu[kk] * e0[kk];
        RC_FOR_K cpd[kk] = cpu[kk] * e0[kk];
    } else {
        RC_FOR_K fxd[kk] = fw[kk] * xd[kk];
        RC_FOR_K cmd[kk] = am2[kk] * fxd[kk];
        RC_FOR_K cpd[kk] = ap2[kk] * fxd[kk];
    }
    RC_FOR_K B0[kk] = fma(gcq, g[kk].q2, 1.0);
    if (PLAIN) {
        RC_FOR_K Aqq[kk] = B0[kk] * A0;
        RC_FOR_K Y[kk] = 0.0;
    } else {
        RC_FOR_K Aqq[kk] = fma(B0[kk], A0, -((c15 * g[kk].u1) * hz[kk]));
        RC_FOR_K Y[kk] = gmc * g[kk].u1;
    }
    RC_FOR_K ee[kk] = fma(EP, et[kk], -1.0);
    RC_FOR_K ff[kk] = fma(-EM, et[kk], 1.0);
    double vp[NA], vn[NA], eo[NA], t2[NA], t1[NA], e0o[NA];
    const double w2A0 = w2pi * A0;
    if (PZ_REFL_DIET) {                                 // reflected_layer, PZ_REFL_DIET (2)
        const double w2gp = w2pi * gp, w2gmc = PLAIN ? 0.0 : w2pi * gmc;
        RC_FOR_K Trd[kk] = st[kk].T * rd[kk];
        RC_FOR_K X[kk] = w2gp * B0[kk];
        RC_FOR_K Y[kk] = PLAIN ? 0.0 : w2gmc * g[kk].u1;
        RC_FOR_K vp[kk] = ((Trd[kk] * (PLAIN ? X[kk] : X[kk] + Y[kk])) * lp1[kk]) * ee[kk];
        RC_FOR_K vn[kk] = ((Trd[kk] * (PLAIN ? X[kk] : X[kk] - Y[kk])) * lm1[kk]) * ff[kk];
    } else {
        RC_FOR_K X[kk] = gp * B0[kk];
        RC_FOR_K Tw[kk] = st[kk].T * w2pi;
        RC_FOR_K Trd[kk] = Tw[kk] * rd[kk];
        RC_FOR_K vp[kk] = (Trd[kk] * lp1[kk]) * ((PLAIN ? X[kk] : X[kk] + Y[kk]) * ee[kk]);
        RC_FOR_K vn[kk] = (Trd[kk] * lm1[kk]) * ((PLAIN ? X[kk] : X[kk] - Y[kk]) * ff[kk]);
    }
    if (!first && eo_ok) {
        RC_FOR_K eo[kk] = st[kk].EO;
    } else {
        RC_FOR_K eo[kk] = fexp2_cold(tauo * (ZP ? g[kk].nl1 : g[kk].nl0), K);
    }
    RC_FOR_K t2[kk] = fma(-e0[kk], et[kk], 1.0);
    if (same_dt) {
        RC_FOR_K t1[kk] = t2[kk];
        RC_FOR_K e0o[kk] = e0[kk];
    } else {
        RC_FOR_K {
            const double e1o = fexp2_cold(dto * g[kk].nl1, K);
            e0o[kk] = ZP ? e1o : fexp2_cold(dto * g[kk].nl0, K);
            t1[kk] = fma(-e0o[kk], e1o, 1.0);
        }
    }
    if (!last) RC_FOR_K st[kk].EO = eo[kk] * e0o[kk];
    double s1[NA], S0[NA], kap[NA], Tn[NA], delta_n[NA];
    if (PZ_REFL_DIET && ZP && nocld && same_dt) {       // reflected_layer, PZ_REFL_DIET (3)
        RC_FOR_K S0[kk] = fma(w2A0, B0[kk] * fx[kk], ssa_h * eo[kk]) * t2[kk];
    } else {
        RC_FOR_K s1[kk] = (ssa_h * (ZP ? 1.0 : g[kk].wq2)) * (eo[kk] * t1[kk]);
        RC_FOR_K S0[kk] = fma((w2pi * (ZP ? 1.0 : g[kk].wq2)) * fx[kk], Aqq[kk] * t2[kk], s1[kk]);
    }
    RC_FOR_K kap[kk] = fma(st[kk].T, S0[kk], st[kk].KAPPA);
    RC_FOR_K Tn[kk] = st[kk].T * et[kk];
    if (last) {                                            // xint[n] = flux_zero/pi (fluxes.py:1266-1270)
        RC_FOR_K vp[kk] = fma(Tn[kk] * EP, 1.0 / PI, vp[kk]);
        RC_FOR_K vn[kk] = fma((Tn[kk] * gam) * EM, 1.0 / PI, vn[kk]);
        RC_FOR_K kap[kk] = fma(Tn[kk] * cpd[kk], 1.0 / PI, kap[kk]);
    }
    if (first) {                                           // top row (fluxes.py:155-158)
        RC_FOR_K delta_n[kk] = b_top - cmu[kk];
        RC_FOR_K st[kk].ZETA = fma(-vn[kk], gam, vp[kk]);
        RC_FOR_K kap[kk] = fma(vn[kk], delta_n[kk], kap[kk]);
    } else {
        double rP[NA], rM[NA], tt[NA];
        RC_FOR_K rP[kk] = cpu[kk] - st[kk].D1;
        RC_FOR_K rM[kk] = cmu[kk] - st[kk].D2;
        RC_FOR_K delta_n[kk] = fma(a2i, rP[kk], -(a1i * rM[kk]));
        RC_FOR_K tt[kk] = fma(gam, delta_n[kk], rP[kk]) * ia;
        RC_FOR_K kap[kk] = fma(st[kk].ZETA, tt[kk], fma(vn[kk], delta_n[kk], kap[kk]));
        RC_FOR_K st[kk].ZETA = fma(-vn[kk], rho_n, fma(st[kk].ZETA, sfac, vp[kk]));
    }
    RC_FOR_K st[kk].XU = xd[kk];
    RC_FOR_K st[kk].KAPPA = kap[kk];
    RC_FOR_K st[kk].T = Tn[kk];
    RC_FOR_K st[kk].D1 = fma(gEM, delta_n[kk], cpd[kk]);
    RC_FOR_K st[kk].D2 = fma(EM, delta_n[kk], cmd[kk]);
    if (last) {
        RC_FOR_K {
            st[kk].l_gam = gam;
            st[kk].l_EM = EM;
            st[kk].l_rho = rho_n;
        }
    }
}

#define RC_SYNC() __syncthreads()

struct RcShared {
    double rho, pgam, pEM;                      // the elimination recurrence of reflected_layer (S.rho, S.pgam, S.pEM)
};

// One layer of wave S: the angle-independent part of reflected_layer<.., FAST = true> (fluxes.py:1132-1141,
// 1172-1177) and the elimination factors shared by all angles.  PLAIN as in rc_angle (then the layer is an interior
// one as well): reflected_layer's NC copy, the cloud terms folded by hand.
template <bool PLAIN>
__device__ __forceinline__ void rc_shared(const ReflectedArgs &a, const double (*in)[64], double (*slot)[64], int fl,
                                          bool not_first, int lane, double F, double cos_theta, const Exp2Coef &K,
                                          RcShared &sh)
{
#pragma clang fp contract(off)
    const double clip = 35.0;                         // fluxes.py:1174
    const double dt = in[RW_DT][lane], w0 = in[RW_W0][lane], fr = in[RW_FR][lane], w0o = in[RW_W0O][lane];
    const bool nocld = PLAIN || (fl & RCF_NOCLD);
    double g1, g2, lam, lam2, fcg = 0.0, ps;
    if (PLAIN) {
        toon_gammas_nocld(0, w0, g1, g2, lam, lam2);
        ps = fr * (0.75 * fma(cos_theta, cos_theta, 1.0));
    } else {
        const double cg = in[RW_G][lane], fc = in[RW_FC][lane];
        fcg = fc * cg;
        toon_gammas(0, w0, fcg, g1, g2, lam, lam2);
        if (nocld)
            ps = fr * (0.75 * fma(cos_theta, cos_theta, 1.0));
        else
            ps = p_single<false>(3, in[RW_CBO][lane], in[RW_GCOS2][lane], fc, fr, cos_theta, a.frac_a, a.frac_b, 2.0,
                                 a.constant_back, a.constant_forward);
    }
    const double E = fmin(lam * dt, clip);
    const double EP = fexp2(E * -NEG_LOG2E, K);
#if PZ_REFL_DIET
    const double r_ge = frcp(g2 * EP);                 // reflected_layer, PZ_REFL_DIET (5)
    const double gam = (g1 - lam) * (r_ge * EP), EM = r_ge * g2;
#else
    const double gam = (g1 - lam) * frcp(g2);
    const double EM = frcp(EP);
#endif
    const double ssa_h = (w0o * F * (0.125 / PI)) * ps;
    const double w2pi = w0 * (0.5 / PI);
    const double Fw0h = (0.5 * F) * w0;
    const double c2 = PLAIN ? 0.0 : SQ3 * fcg;
    const double A0 = PLAIN ? (g1 + g2) : (g1 + g2) + c2;
    double a1i = 0.0, a2i = 0.0, ia = 0.0, rho_n = gam, sfac = 0.0;
    if (PLAIN || not_first) {
        const double em2 = sh.pEM * sh.pEM;
        const double a1 = fma(-(sh.pgam * em2), sh.rho, 1.0);
        const double a2 = fma(-em2, sh.rho, sh.pgam);
        const double d1 = fma(-gam, a2, a1);
        const double r12 = frcp(d1 * a1);
        const double inv = r12 * a1;
        a1i = a1 * inv;
        a2i = a2 * inv;
        rho_n = fma(gam, a1i, -a2i);
        ia = sh.pEM * (r12 * d1);
        sfac = fma(-gam, rho_n, 1.0) * ia;
    }
    sh.rho = rho_n;
    sh.pgam = gam;
    sh.pEM = EM;
    slot[RC_LAM][lane] = lam;
    slot[RC_EP][lane] = EP;
    slot[RC_EM][lane] = EM;
    slot[RC_GAM][lane] = gam;
    slot[RC_FW0H][lane] = Fw0h;
    slot[RC_A0][lane] = A0;
    slot[RC_W2PI][lane] = w2pi;
    slot[RC_SSAH][lane] = ssa_h;
    slot[RC_A1I][lane] = a1i;
    slot[RC_A2I][lane] = a2i;
    slot[RC_IA][lane] = ia;
    slot[RC_SFAC][lane] = sfac;
    slot[RC_RHON][lane] = rho_n;
    if (!PLAIN && !nocld) {
        slot[RC_A1][lane] = c2 * (g1 - g2);
        const double c15 = 1.5 * fcg;
        slot[RC_C15][lane] = c15;
        slot[RC_GMC][lane] = (1.0 - gam) * c15;
    }
}

// NL-wide fsqrt (device_math.hpp), as fexp2_n / frcp_n above
template <int NL>
__device__ __forceinline__ void fsqrt_n(const double (&x)[NL], double (&out)[NL])
{
#pragma clang fp contract(off)
    double y[NL], g[NL], h[NL], r[NL], d[NL];
#define RC_FOR_J _Pragma("unroll") for (int j = 0; j < NL; ++j)
    RC_FOR_J y[j] = __builtin_amdgcn_rsq(x[j]);
    RC_FOR_J g[j] = x[j] * y[j];
    RC_FOR_J h[j] = 0.5 * y[j];
    RC_FOR_J r[j] = fma(-h[j], g[j], 0.5);
    RC_FOR_J g[j] = fma(g[j], r[j], g[j]);
    RC_FOR_J h[j] = fma(h[j], r[j], h[j]);
    RC_FOR_J r[j] = fma(-h[j], g[j], 0.5);
    RC_FOR_J g[j] = fma(g[j], r[j], g[j]);
    RC_FOR_J h[j] = fma(h[j], r[j], h[j]);
    RC_FOR_J d[j] = fma(-g[j], g[j], x[j]);
    RC_FOR_J g[j] = fma(d[j], h[j], g[j]);
    RC_FOR_J out[j] = (x[j] == 0.0) ? x[j] : g[j];
}

// A whole PLAIN round of wave S (NL interior layers without cloud, all shortcuts): rc_shared<true> for each of
// them, with the layer-local chains (gammas -> square root -> Gamma; lambda dtau -> exp -> reciprocal) written over
// the NL layers so that they fill each other's latency -- alone they are ~45 dependent instructions per layer, which
// made wave S the slowest wave of the workgroup -- and only the rho recurrence left sequential.
template <int NL>
__device__ __forceinline__ void rc_shared_plain_round(const double (*in)[RW_NV][64], double (*slot)[RC_NV][64],
                                                      int lane, double F, double cos_theta, const Exp2Coef &K,
                                                      RcShared &sh)
{
#pragma clang fp contract(off)
    double dt[NL], w0[NL], fr[NL], w0o[NL];
    RC_FOR_J {
        dt[j] = in[j][RW_DT][lane];
        w0[j] = in[j][RW_W0][lane];
        fr[j] = in[j][RW_FR][lane];
        w0o[j] = in[j][RW_W0O][lane];
    }
    double g1[NL], g2[NL], aa[NL], bb[NL], dd[NL], lam[NL], ig2[NL], gam[NL], E[NL], targ[NL], EP[NL], EM[NL];
    RC_FOR_J g1[j] = (SQ3 * 0.5) * (2.0 - w0[j]);               // toon_gammas_nocld, quadrature
    RC_FOR_J g2[j] = SQ3 * w0[j] * 0.5;
    RC_FOR_J aa[j] = g1[j] * g1[j];
    RC_FOR_J bb[j] = g2[j] * g2[j];
    RC_FOR_J dd[j] = aa[j] - bb[j];
    fsqrt_n<NL>(dd, lam);
    RC_FOR_J E[j] = fmin(lam[j] * dt[j], 35.0);                 // fluxes.py:1174
    RC_FOR_J targ[j] = E[j] * -NEG_LOG2E;
    fexp2_n<NL>(targ, K, EP);
#if PZ_REFL_DIET
    {                                                           // reflected_layer, PZ_REFL_DIET (5)
        double pge[NL], rge[NL];
        RC_FOR_J pge[j] = g2[j] * EP[j];
        frcp_n<NL>(pge, rge);
        RC_FOR_J ig2[j] = rge[j] * EP[j];
        RC_FOR_J EM[j] = rge[j] * g2[j];
    }
    RC_FOR_J gam[j] = (g1[j] - lam[j]) * ig2[j];
#else
    frcp_n<NL>(g2, ig2);
    RC_FOR_J gam[j] = (g1[j] - lam[j]) * ig2[j];
    frcp_n<NL>(EP, EM);
#endif
    double ps[NL], ssa_h[NL], w2pi[NL], Fw0h[NL], A0[NL];
    RC_FOR_J ps[j] = fr[j] * (0.75 * fma(cos_theta, cos_theta, 1.0));
    RC_FOR_J ssa_h[j] = (w0o[j] * F * (0.125 / PI)) * ps[j];
    RC_FOR_J w2pi[j] = w0[j] * (0.5 / PI);
    RC_FOR_J Fw0h[j] = (0.5 * F) * w0[j];
    RC_FOR_J A0[j] = g1[j] + g2[j];
    RC_FOR_J {
        slot[j][RC_LAM][lane] = lam[j];
        slot[j][RC_EP][lane] = EP[j];
        slot[j][RC_EM][lane] = EM[j];
        slot[j][RC_GAM][lane] = gam[j];
        slot[j][RC_FW0H][lane] = Fw0h[j];
        slot[j][RC_A0][lane] = A0[j];
        slot[j][RC_W2PI][lane] = w2pi[j];
        slot[j][RC_SSAH][lane] = ssa_h[j];
    }
    RC_FOR_J {                                                  // the one sequential part: the recurrence rho
        const double em2 = sh.pEM * sh.pEM;
        const double a1 = fma(-(sh.pgam * em2), sh.rho, 1.0);
        const double a2 = fma(-em2, sh.rho, sh.pgam);
        const double d1 = fma(-gam[j], a2, a1);
        const double r12 = frcp(d1 * a1);
        const double inv = r12 * a1;
        const double a1i = a1 * inv;
        const double a2i = a2 * inv;
        const double rho_n = fma(gam[j], a1i, -a2i);
        const double ia = sh.pEM * (r12 * d1);
        const double sfac = fma(-gam[j], rho_n, 1.0) * ia;
        sh.rho = rho_n;
        sh.pgam = gam[j];
        sh.pEM = EM[j];
        slot[j][RC_A1I][lane] = a1i;
        slot[j][RC_A2I][lane] = a2i;
        slot[j][RC_IA][lane] = ia;
        slot[j][RC_SFAC][lane] = sfac;
        slot[j][RC_RHON][lane] = rho_n;
    }
#undef RC_FOR_J
}

// ZP: every angle has ubar0 == ubar1 (zero phase angle), as in k_reflected_toa.
// DRV: which planes exist -- 0 all eleven; 2 only dtau and w0 (an atmosphere without cloud: the others are constants,
// copies and running sums); 3 everything but tau, tau_og and gcos2 (running sums and 0.5 ftau_ray) -- the two patterns
// picaso() / the C driver hand over when picaso_reflected_1d_can_derive says yes (k_reflected_toa's DRV = 2 / 3).  Wave L
// forms the missing values from the ones it loaded with the operations compute_opacity uses (optics.py:342, 353-354,
// 412-420), so the LDS ring holds the same eleven numbers per layer as with all planes in HBM: same bits downstream.
template <bool ZP, int DRV>
__global__ __launch_bounds__(64 * (RC_MAX_ANGLES / RC_APW + 2)) void k_reflected_coop(const ReflectedArgs a)
{
#pragma clang fp contract(off)      // operations as written: the same ones reflected_layer performs
    constexpr bool CLEAR = (DRV == 2), LEVELS = (DRV == 3), NOSUMS = CLEAR || LEVELS;
    __shared__ double raw[3][RC_R][RW_NV][64];        // plane values, two rounds ahead of the angle waves
    __shared__ double ring[2][RC_R][RC_NV][64];       // wave S's results, one round ahead
    __shared__ int lflag[3][RC_R];                    // wave-uniform shortcut flags of a layer (wave L)
    __shared__ int rplain[3];                         // 1: every layer of the round is interior with all flags set
    __shared__ double xs[RC_MAX_ANGLES][64];
    const int lane = threadIdx.x & 63;
    const int hw_wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nang = a.na;
    const int nawaves = (nang + RC_APW - 1) / RC_APW; // angle waves; the workgroup has nawaves + 2 waves
    const int lwave = nawaves + 1;                    // the last wave loads
    const int swave = (PZ_RCOOP_SWAVE < lwave) ? PZ_RCOOP_SWAVE : 0;
    const bool is_l = hw_wave == lwave, is_s = hw_wave == swave;
    const int aw = (is_s || is_l) ? 0 : (hw_wave < swave ? hw_wave : hw_wave - 1);    // index of an angle wave
    const int k0 = aw * RC_APW;                       // its first angle
    long col = (long)blockIdx.x * 64 + lane;
    const bool active = col < a.ncol;
    if (!active) col = a.ncol - 1;                    // padding lanes shadow the last column, never store
    const long w = a.ncolper > 1 ? col / a.ncolper : col;
    const int n = a.nlayer;
    const long pitch = a.pitch;
    const int nrounds = (n + RC_R - 1) / RC_R;
    // Schedule (steps separated by one workgroup barrier): in step t wave L loads the planes of round t + 2, wave S
    // turns the planes of round t + 1 into the shared layer quantities, the angle waves consume round t.

    if (is_l) {
        // ------------------------------------------------------------------------------------------------
        // wave L: HBM -> LDS, every plane value read exactly once, and the wave-uniform shortcut tests of
        // k_reflected_toa's `prep` (bit-exact comparisons, see reflected_layer) on the values it holds
        // ------------------------------------------------------------------------------------------------
        // RW_DT, RW_TAUN, RW_W0, RW_G, RW_GCOS2, RW_FC, RW_FR, RW_DTO, RW_TAUO, RW_W0O, RW_CBO: which are in HBM
        constexpr bool have[RW_NV] = {true, !NOSUMS, true, !CLEAR, !NOSUMS, !CLEAR, !CLEAR, !CLEAR, !NOSUMS, !CLEAR, !CLEAR};
        const double *pl[RW_NV] = {a.dtau + col, have[RW_TAUN] ? a.tau + col + pitch : nullptr, a.w0 + col,
                                   have[RW_G] ? a.cosb + col : nullptr, have[RW_GCOS2] ? a.gcos2 + col : nullptr,
                                   have[RW_FC] ? a.ftau_cld + col : nullptr, have[RW_FR] ? a.ftau_ray + col : nullptr,
                                   have[RW_DTO] ? a.dtau_og + col : nullptr, have[RW_TAUO] ? a.tau_og + col : nullptr,
                                   have[RW_W0O] ? a.w0_og + col : nullptr, have[RW_CBO] ? a.cosb_og + col : nullptr};
        double tau_i = NOSUMS ? 0.0 : a.tau[col];
        double tauo_pred = 0.0;                       // tau_og[i-1] + dtau_og[i-1] of the layer above
        // the values of the planes that are not in HBM, from the ones that are (in layer order: the running sums)
        auto derive = [&](double (&x)[RW_NV]) {
            if (CLEAR) {
                x[RW_G] = 0.0; x[RW_FC] = 0.0; x[RW_FR] = 1.0; x[RW_GCOS2] = 0.5; x[RW_CBO] = 0.0;
                x[RW_DTO] = x[RW_DT]; x[RW_W0O] = x[RW_W0];
            } else if (LEVELS) {
                x[RW_GCOS2] = 0.5 * x[RW_FR];
            }
            if (NOSUMS) { x[RW_TAUN] = tau_i + x[RW_DT]; x[RW_TAUO] = tauo_pred; }
        };
        // the loads of round q + 1 are issued BEFORE round q's values (loaded a step earlier) are written to LDS and
        // the step's barrier is reached: a round's loads stay in flight for a whole step instead of holding it up
        double v[2][RC_R][RW_NV];
        auto issue = [&](int q, double (&dst)[RC_R][RW_NV]) {
#pragma unroll
            for (int j = 0; j < RC_R; ++j) {
                const int i = q * RC_R + j;
                if (i < n) {
                    const long o = (long)i * pitch;
#pragma unroll
                    for (int p = 0; p < RW_NV; ++p)
                        if (have[p]) dst[j][p] = pl[p][o];
                }
            }
        };
        auto layer_flags = [&](const double (&x)[RW_NV]) {
            const bool cum_tau = __all(x[RW_TAUN] == tau_i + x[RW_DT]);
            const bool eo_ok = __all(x[RW_TAUO] == tauo_pred);
            const bool same_dt = __all(x[RW_DTO] == x[RW_DT]);
            const bool nocld = __all(x[RW_FC] == 0.0);
            tau_i = x[RW_TAUN];
            tauo_pred = x[RW_TAUO] + x[RW_DTO];
            return (cum_tau ? RCF_CUM_TAU : 0) | (eo_ok ? RCF_EO_OK : 0) | (same_dt ? RCF_SAME_DT : 0) |
                   (nocld ? RCF_NOCLD : 0);
        };
        auto store = [&](int q, const double (&src)[RC_R][RW_NV]) {
            int all = RCF_ALL;
#pragma unroll
            for (int j = 0; j < RC_R; ++j) {
                const int i = q * RC_R + j;
                if (i < n) {
                    double x[RW_NV];
#pragma unroll
                    for (int p = 0; p < RW_NV; ++p) x[p] = have[p] ? src[j][p] : 0.0;
                    derive(x);
#pragma unroll
                    for (int p = 0; p < RW_NV; ++p) raw[q % 3][j][p][lane] = x[p];
                    const int fl = layer_flags(x);
                    if (lane == 0) lflag[q % 3][j] = fl;
                    all &= (i > 0 && i < n - 1) ? fl : 0;
                } else {
                    all = 0;
                }
            }
            if (lane == 0) rplain[q % 3] = (all == RCF_ALL);
        };
        // Steady state without a branch around the loads: a conditional issue makes the compiler's wait-count
        // bookkeeping give up at the join (s_waitcnt vmcnt(0) before every reissue: the loads of a step then no
        // longer stay in flight across its barrier), so full rounds run in a loop of their own.
        auto issue_full = [&](int q, double (&dst)[RC_R][RW_NV]) {
#pragma unroll
            for (int j = 0; j < RC_R; ++j) {
                const long o = (long)(q * RC_R + j) * pitch;
#pragma unroll
                for (int p = 0; p < RW_NV; ++p)
                    if (have[p]) dst[j][p] = pl[p][o];
            }
        };
        auto store_full = [&](int q, const double (&src)[RC_R][RW_NV]) {     // 0 < layers < n - 1
            int all = RCF_ALL;
#pragma unroll
            for (int j = 0; j < RC_R; ++j) {
                double x[RW_NV];
#pragma unroll
                for (int p = 0; p < RW_NV; ++p) x[p] = have[p] ? src[j][p] : 0.0;
                derive(x);
#pragma unroll
                for (int p = 0; p < RW_NV; ++p) raw[q % 3][j][p][lane] = x[p];
                const int fl = layer_flags(x);
                if (lane == 0) lflag[q % 3][j] = fl;
                all &= fl;
            }
            if (lane == 0) rplain[q % 3] = (all == RCF_ALL);
        };
        const int nfull = n / RC_R;                   // rounds with all RC_R layers
        const int nsteps = nrounds + 2;
        // step s (s = 0, 1: the two prologue steps; then the nrounds main steps) stores round s, issues round s + 1
        issue(0, v[0]);
        int s2 = 0;
        if (nfull > 3) {                              // round 0 holds the first layer: general form
            issue_full(1, v[1]);
            store(0, v[0]);
            RC_SYNC();
            issue_full(2, v[0]);
            store_full(1, v[1]);
            RC_SYNC();
            s2 = 2;
            for (; s2 + 3 < nfull; s2 += 2) {         // rounds s2 .. s2 + 2 are full and end before the last layer
                issue_full(s2 + 1, v[1]);             // unrolled by two: the register sets alternate without copies
                store_full(s2, v[0]);
                RC_SYNC();
                issue_full(s2 + 2, v[0]);
                store_full(s2 + 1, v[1]);
                RC_SYNC();
            }
        }
        for (; s2 < nsteps; s2 += 2) {
            if (s2 + 1 < nrounds) issue(s2 + 1, v[1]);
            if (s2 < nrounds) store(s2, v[0]);
            RC_SYNC();
            if (s2 + 1 < nsteps) {
                if (s2 + 2 < nrounds) issue(s2 + 2, v[0]);
                if (s2 + 1 < nrounds) store(s2 + 1, v[1]);
                RC_SYNC();
            }
        }
        RC_SYNC();
        return;
    }

    const double b_top = a.b_top;
    const double cos_theta = ZP ? 1.0 : a.cos_theta;
    const double F = a.F0PI[w], rs = a.surf_reflect[w];
    Exp2Coef K;
    K.load();

    if (is_s) {
        // ------------------------------------------------------------------------------------------------
        // wave S: one round ahead of the angle waves
        // ------------------------------------------------------------------------------------------------
        RcShared sh{0.0, 0.0, 0.0};
        auto produce_round = [&](int q) {
            const int plain = __builtin_amdgcn_readfirstlane(rplain[q % 3]);
            if (plain) {                              // straight-line, the layers of the round interleaved
                rc_shared_plain_round<RC_R>(raw[q % 3], ring[q & 1], lane, F, cos_theta, K, sh);
            } else {
#pragma unroll
                for (int j = 0; j < RC_R; ++j) {
                    const int i = q * RC_R + j;
                    if (i < n) {
                        const int fl = __builtin_amdgcn_readfirstlane(lflag[q % 3][j]);
                        rc_shared<false>(a, raw[q % 3][j], ring[q & 1][j], fl, i > 0, lane, F, cos_theta, K, sh);
                    }
                }
            }
        };
        RC_SYNC();                              // round 0 of the planes is in LDS
        produce_round(0);
        RC_SYNC();
        for (int t = 0; t < nrounds; ++t) {
            if (t + 1 < nrounds) produce_round(t + 1);
            RC_SYNC();
        }
        // fused disco.compress_disco (disco.py:145-148): one running sum over the angles in (g, t) order
        RC_SYNC();
        if (a.albedo && active) {
            double alb = a.albedo_first ? 0.0 : a.albedo[w];
            for (int j = 0; j < nang; ++j) alb = alb + xs[j][lane] * a.ang[j].wgt * a.ang[j].wgt2;
            if (a.albedo_last) alb = a.albedo_scale * alb / F * (a.cos_theta + 1.0);
            a.albedo[w] = alb;
        }
        return;
    }

    // ----------------------------------------------------------------------------------------------------
    // wave A_k: the per-angle part of the sweep (reflected_layer's angle loop body, iteration k)
    // ----------------------------------------------------------------------------------------------------
    auto angle_wave = [&](auto na_c) {
        constexpr int NA = decltype(na_c)::value;
        ReflectedArgs::Angle g[NA];
        RcState st[NA];
#pragma unroll
        for (int kk = 0; kk < NA; ++kk) {
            g[kk] = a.ang[k0 + kk];
            st[kk].T = 1.0;
            st[kk].XU = fexp2(mul_unfused(NOSUMS ? 0.0 : a.tau[col], ZP ? g[kk].nl1 : g[kk].nl0), K);
            st[kk].EO = st[kk].KAPPA = st[kk].ZETA = st[kk].D1 = st[kk].D2 = 0.0;
            st[kk].l_gam = st[kk].l_EM = st[kk].l_rho = 0.0;
        }
        auto consume_round = [&](int q, int plain) {
            if (plain) {                              // interior layers, all shortcuts, no cloud: straight-line
#pragma unroll
                for (int j = 0; j < RC_R; ++j)
                    rc_angle<ZP, false, false, true, NA>(raw[q % 3][j], ring[q & 1][j], RCF_ALL, lane, g, K, b_top, st);
                return;
            }
#pragma unroll
            for (int j = 0; j < RC_R; ++j) {
                const int i = q * RC_R + j;
                if (i >= n) break;
                const double(*in)[64] = raw[q % 3][j];
                const double(*slot)[64] = ring[q & 1][j];
                const int fl = __builtin_amdgcn_readfirstlane(lflag[q % 3][j]);
                if (i == 0) {
                    if (n == 1) rc_angle<ZP, true, true, false, NA>(in, slot, fl, lane, g, K, b_top, st);
                    else rc_angle<ZP, true, false, false, NA>(in, slot, fl, lane, g, K, b_top, st);
                } else if (i == n - 1) {
                    rc_angle<ZP, false, true, false, NA>(in, slot, fl, lane, g, K, b_top, st);
                } else {
                    rc_angle<ZP, false, false, false, NA>(in, slot, fl, lane, g, K, b_top, st);
                }
            }
        };
        RC_SYNC();                                    // planes of round 0 (and its flags)
        int plain = __builtin_amdgcn_readfirstlane(rplain[0]);
        RC_SYNC();                                    // wave S's round 0; planes of round 1
        for (int t = 0; t < nrounds; ++t) {
            // the flags of the next round are in LDS already (wave L is two rounds ahead): read them before this
            // round's arithmetic, so that the step does not begin with an exposed LDS round trip
            const int plain_next = (t + 1 < nrounds) ? __builtin_amdgcn_readfirstlane(rplain[(t + 1) % 3]) : 0;
            consume_round(t, plain);
            plain = plain_next;
            RC_SYNC();
        }
        // ---- surface row (fluxes.py:178-183) and output: the epilogue of k_reflected_toa for these angles ----
#pragma unroll
        for (int kk = 0; kk < NA; ++kk) {
            const double em2 = st[kk].l_EM * st[kk].l_EM;
            const double bden = 1.0 / fma(-(em2 * (st[kk].l_gam - rs)), st[kk].l_rho, fma(-rs, st[kk].l_gam, 1.0));
            const double b_surface = ((rs * (ZP ? g[kk].u1 : g[kk].u0)) * F) * st[kk].XU;
            const double pos = (st[kk].l_EM * fma(rs, st[kk].D2, b_surface - st[kk].D1)) * bden;
            const double x = fma(st[kk].ZETA, pos, st[kk].KAPPA);
            if (active) a.xint[(long)(k0 + kk) * a.ncol + col] = x;
            xs[k0 + kk][lane] = x;
        }
        RC_SYNC();
    };
    static_assert(RC_APW == 1 || RC_APW == 2, "angle waves carry one or two angles");
    if (RC_APW == 2 && nang - k0 >= 2) angle_wave(std::integral_constant<int, RC_APW>{});
    else angle_wave(std::integral_constant<int, 1>{});
}

// 0: all eleven planes; 2: only dtau and w0; 3: all but tau, tau_og, gcos2; -1: another pattern (not for this kernel)
int reflected_coop_pattern(const ReflectedArgs &a)
{
    const bool cloudset = a.cosb && a.cosb_og && a.ftau_cld && a.ftau_ray, og = a.dtau_og && a.w0_og;
    if (a.tau && a.tau_og && a.gcos2 && cloudset && og) return 0;
    if (!a.tau && !a.tau_og && !a.gcos2 && cloudset && og) return 3;
    if (!a.tau && !a.tau_og && !a.gcos2 && !a.cosb && !a.cosb_og && !a.ftau_cld && !a.ftau_ray && !a.dtau_og && !a.w0_og) return 2;
    return -1;
}

bool reflected_coop_ok(const ReflectedArgs &a)
{
    if (reflected_coop_pattern(a) < 0) return false;
    if (getenv("PICASO_AMD_REFL_NO_COOP")) return false;
    if (a.na < 1 || a.na > RC_MAX_ANGLES || a.ny > 1 || a.nlayer < 1) return false;
    // the reference's default options (what reflected_layer<.., FAST = true> fixes at compile time)
    return a.toon_coefficients == 0 && a.single_phase == 3 && a.multi_phase == 0 && a.frac_c == 2.0;
}


int launch_reflected_coop(picaso_ctx *ctx, const ReflectedArgs &a)
{
    if (a.ncol <= 0) return fail(ctx, "reflected: empty problem");
    bool zp = true;
    for (int k = 0; k < a.na; ++k) zp = zp && (a.ang[k].u0 == a.ang[k].u1);
    if (zp && a.cos_theta != 1.0) zp = false;          // as fast_options: the ZP variant fixes cos_theta = 1
    const dim3 grid((unsigned)((a.ncol + 63) / 64)), block(64 * ((a.na + RC_APW - 1) / RC_APW + 2));
    const int drv = reflected_coop_pattern(a);
    if (drv < 0) return fail(ctx, "reflected (cooperative kernel): unsupported set of planes");
#define PZ_GO(Z, D) hipLaunchKernelGGL((k_reflected_coop<Z, D>), grid, block, 0, ctx->stream, a)
    if (zp) { if (drv == 0) PZ_GO(true, 0); else if (drv == 2) PZ_GO(true, 2); else PZ_GO(true, 3); }
    else { if (drv == 0) PZ_GO(false, 0); else if (drv == 2) PZ_GO(false, 2); else PZ_GO(false, 3); }
#undef PZ_GO
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace pz
