// Toon89 two-stream reflected light, top-of-atmosphere intensity -- gfx950.
//
// Replaces fluxes.get_reflected_1d / get_reflected_3d (reference picaso/fluxes.py:1009-1413,
// :354-660) for the spectrum path (get_toa_intensity=1, get_lvl_flux=0).
//
// Mapping: one LANE per wavelength column (3-D: per (wavelength, facet) column, facet index
// fastest exactly as the reference stores it), so every plane read is a coalesced 512-B row
// segment of the reference's own (nlayer, nwno[, ng, nt]) layout -- no transpose anywhere.
// All `NA` disk angles that share the planes are carried in registers of the same lane: the
// angle-independent work (g1, g2, lambda, Gamma, exp(lambda dtau), p_single, the elimination
// factors) is done once per layer instead of once per angle.
//
// Algorithm: the reference builds a 2n x 2n tridiagonal system per (wavelength, angle), solves it
// (Thomas, two sweeps) and then runs a bottom-up source-function recursion.  The matrix does not
// depend on the angle (only the right-hand side does) and the TOA intensity is a LINEAR functional
// of the solution, so the whole thing collapses into ONE top-down sweep with O(1) state:
//   * unknowns per layer are Toon's own (pos_i, neg_i) = (Y1+Y2, Y1-Y2); the flux-continuity
//     equations at interface i|i+1 are (reference fluxes.py:1227-1231 expressions)
//         EP_i pos_i + G_i EM_i neg_i + c+dn_i = pos_{i+1} + G_{i+1} neg_{i+1} + c+up_{i+1}
//         G_i EP_i pos_i + EM_i neg_i + c-dn_i = G_{i+1} pos_{i+1} + neg_{i+1} + c-up_{i+1}
//     (the reference's rows 2i+1, 2i+2, fluxes.py:161-175, are invertible combinations of these);
//   * sweeping down we keep the one relation everything above imposes on layer i,
//         neg_i = delta_i - rho_i pos_i            (rho shared by all angles, delta per angle)
//     and the TOA functional of layers 0..i as an affine function of the only free unknown,
//         J_{<=i} = kappa_i + zeta_i pos_i        (per angle),
//     using pos_{i-1} = s_i pos_i + t_i (contracting: s ~ exp(-lambda dtau));
//   * the surface row (fluxes.py:178-181) fixes pos_{n-1}; xint_at_top = kappa + zeta pos.
// Each input element is read exactly once; nothing but the result is written.  Against the
// reference the result differs only by rounding (<= 1e-10 relative on the golden scenes, at the
// level of the reference's own fp64 conditioning); see tests/test_parity_gpu.py.
//
// The kernel is FP64-VALU bound (~20 fp64 instructions per algorithmic byte at 5 angles), so the
// code below is organised around instruction count: lean exp / reciprocal / rsqrt helpers
// (device_math.hpp), first and last layer peeled out of the loop, host-precomputed per-angle
// constants (SGPR resident), and -- for the symmetric geometry -- exponentials carried as running
// products when the cumulative optical depth really is the running sum of the layer depths.
#include "common.hpp"
#include "device_math.hpp"

namespace pz {

// Two waves per SIMD: one fp64 wave alone issues at most every ~5.1 cycles (tools/ubench/
// f64_rates.hip) while the pipe accepts one per ~3.5-4, and a 1e5-column spectrum is only 1.5 waves
// per SIMD, so co-residency beats the extra scratch spills (measured 0.405 ms vs 0.481 ms).
#ifndef PZ_REFL_MINWAVES
#define PZ_REFL_MINWAVES 2
#endif
// threads per block (a tuning knob of tools/sweep.sh; the LDS state tile is [var][angle][PZ_REFL_BLOCK])
#ifndef PZ_REFL_BLOCK
#define PZ_REFL_BLOCK 256
#endif

// Per-angle sweep state (7 doubles per angle).  D1 = c+dn + Gamma EM delta, D2 = c-dn + EM delta of
// the layer above: the only combinations of (c+dn, c-dn, delta) the next interface and the surface
// row need.  For NA >= 4 part of the state lives in LDS ([var][angle][lane], so a wave's
// ds_read_b64 / ds_write_b64 hit 64 consecutive 8-byte words: conflict free): all in registers the
// 5-angle kernel needs ~300 VGPRs, i.e. either one wave per SIMD or ~50 registers spilled to
// scratch, whose reloads share the vmcnt queue with the plane prefetch.
constexpr int NSTATE = 7;
enum { S_T = 0, S_XU, S_EO, S_KAPPA, S_ZETA, S_D1, S_D2 };   // late-use variables last

// NLDS = how many of the state variables (counted from the END of the enum order below) live in
// LDS when LDS is on; the rest stay in registers.
// Measured on the 5-angle headline case: 0 -> 50 VGPRs spilled to scratch (0.416 ms), 4 -> no
// spills, 41 KB LDS per block (0.410 ms), 7 -> 72 KB (0.421 ms): the kernel is VALU-issue bound,
// so 4 is chosen for being spill-free with the smaller LDS footprint.
#ifndef PZ_REFL_NLDS
#define PZ_REFL_NLDS 4
#endif
// one reciprocal for all angles of a layer (batch inversion) instead of one per angle
// FAST kernels address the planes as SGPR base + 32-bit lane offset (the launcher checks that a plane
// is smaller than 4 GB); 0 = 64-bit lane addresses everywhere (A/B switch)
#ifndef PZ_REFL_SADDR
#define PZ_REFL_SADDR 1
#endif
// second copy of the interior layer body for layers without cloud (FAST kernels only)
#ifndef PZ_REFL_FAST_NONZP
#define PZ_REFL_FAST_NONZP 1
#endif
#ifndef PZ_REFL_LEVELS_VARIANT
#define PZ_REFL_LEVELS_VARIANT 1
#endif
#ifndef PZ_REFL_CLEAR_VARIANT
#define PZ_REFL_CLEAR_VARIANT 1
#endif
#ifndef PZ_REFL_CLOUD_BODY
#define PZ_REFL_CLOUD_BODY 1
#endif
#ifndef PZ_REFL_NOCLD_BODY
#define PZ_REFL_NOCLD_BODY 1
#endif
// One reciprocal for all angles of a layer (Montgomery batch inversion) instead of one per angle:
// -1.3 % time at five angles, but an angle's result then depends (in its last bits) on which angles
// share its lane, i.e. on the angle chunking and on whether a small wavelength shard takes the
// one-angle-per-wave path.  Off: a column's result is bit-identical however the grid is sharded and
// the angles are grouped, which the multi-GPU path and its tests rely on.
#ifndef PZ_REFL_BATCH_RCP
#define PZ_REFL_BATCH_RCP 0
#endif
// PZ_REFL_DIET (common.hpp): the round-5 regrouping of this body's arithmetic; 0 = the round-4 operation order
template <int NA, bool LDS>
struct ReflState {
    static constexpr int NL = LDS ? PZ_REFL_NLDS : 0;      // variables [NSTATE-NL, NSTATE) in LDS
    static constexpr int NR = NSTATE - NL;
    double reg[NR > 0 ? NR : 1][NA];
    double *lds;                                   // this lane's column of the block's LDS tile
    double rho, pgam, pEM;
    __device__ __forceinline__ double get(int var, int k) const
    {
        return (var >= NR) ? lds[((var - NR) * NA + k) * PZ_REFL_BLOCK] : reg[var < NR ? var : 0][k];
    }
    __device__ __forceinline__ void set(int var, int k, double v)
    {
        if (var >= NR) lds[((var - NR) * NA + k) * PZ_REFL_BLOCK] = v;
        else reg[var < NR ? var : 0][k] = v;
    }
};

struct LayerIn {
    double dt, tau_n, w0, g, gcos2, fc, fr, dto, tauo, w0o, cbo;
    // wave-uniform flags, set when the layer is consumed:
    bool cum_tau;    // tau[i+1] == tau[i] + dtau[i] bit-exactly: exp(-tau[i+1]/u) = exp(-tau[i]/u) exp(-dtau/u)
    bool eo_ok;      // tau_og[i] == tau_og[i-1] + dtau_og[i-1]: the carried product is exp(-tau_og[i]/u)
    bool same_dt;    // dtau_og == dtau (no delta-scaling in this layer)
    bool nocld;      // ftau_cld == 0 (no cloud in this layer)
    bool allf;       // cum_tau && eo_ok && same_dt
    bool af_cloud;   // cum_tau && eo_ok && !same_dt on a layer with cloud
};

// One layer of the sweep.  FIRST / LAST are compile-time so the top row and the bottom-boundary
// terms cost nothing inside the loop.  ZP: every angle has ubar0 == ubar1 (symmetric 1-D geometry,
// reference justdoit.py:1513-1532): exp(-dtau (u0+u1)/(u0 u1)) = exp(-dtau/u1)^2, and when the
// level optical depths are the exact running sums of the layer depths (how compute_opacity builds
// them, optics.py:353-354, 418-420) exp(-tau[i+1]/u0) = exp(-tau[i]/u0) exp(-dtau[i]/u1) needs no
// exponential of its own.  The check is bit-exact and per wave, so arbitrary caller-supplied tau
// planes still take the direct exp(-tau/u0).
// FAST: the reference's default options (config.json: quadrature coefficients, TTHG_ray single
// scattering with frac_c = 2, N=2 multiple scattering) in the symmetric zero-phase geometry
// (cos_theta = 1), fixed at compile time: no option branches in the layer loop and the option values
// leave the SGPR file (the generic 5-angle kernel spills 49 SGPRs to VGPR lanes, this one 23).
// NC: no cloud anywhere in this layer of the wave (ftau_cld == 0 in every lane, checked per wave by
// the caller): every term that carries ftau_cld*cosb drops out.  The remaining operations are the
// generic ones with fcg = +0 folded by hand (x + 0, x * 1 and fma(0, y, z) are exact), so the result
// is bit-identical to the generic body on such a layer.
// AF: the three wave-uniform shortcuts (cum_tau, eo_ok, same_dt) all hold on this layer, known at compile time
// in this copy of the body: no per-angle branches and no fall-back exponentials in the instruction stream.  The
// checks themselves cost nothing; the 3-4 wave-uniform branches per angle they fed did -- with the flags assumed
// the five-angle launch ran 14 % faster and a one-wave-per-SIMD launch 20-27 % (branch bubbles with nothing to
// hide them, basic blocks too small to schedule across).  Same operations as the general body takes when
// the flags are true, so the bits do not change.
// SDT (with AF): 1 = same_dt holds as well (the cloud-free copy), 2 = it does not (the usual cloud layer: delta-
// scaled, so dtau_og != dtau and the two extra exponentials are always formed) -- both straight-line.
template <int NA, bool IS3D, bool ZP, bool FIRST, bool LAST, bool LDS, bool FAST = false, bool NC = false,
          bool AF = false, int SDT = 1>
__device__ __forceinline__ void reflected_layer(const ReflectedArgs &a, const LayerIn &L,
                                                ReflState<NA, LDS> &S,
                                                const ReflectedArgs::Angle (&g)[NA],
                                                const Exp2Coef &K, double F, double clip, int tc_,
                                                double b_top)
{
    // Every floating-point operation below is written out: contraction is off and each fused
    // multiply-add is an explicit fma().  With the compiler free to contract (hipcc's default) the
    // template instantiations (angles per lane, FAST, NC, first/last layer) each got their own choice
    // of fused operations and disagreed in the last bits (5e-12 relative between the one-angle and the
    // five-angle kernel): a column's result depended on how the launch was shaped.  Now it does not.
#pragma clang fp contract(off)
    const int tc = FAST ? 0 : tc_;
    const int single_phase = FAST ? 3 : a.single_phase;
    const int multi_phase = FAST ? 0 : a.multi_phase;
    const double cos_theta = (FAST && ZP) ? 1.0 : a.cos_theta;   // zero phase: cos_theta = 1 (what ZP geometry means)
    const double frac_c = FAST ? 2.0 : a.frac_c;
    const double dt = L.dt, w0 = L.w0;
    // ---- angle-independent layer quantities (fluxes.py:1132-1141, 1172-1177) ----
    const double fcg = NC ? 0.0 : L.fc * L.g;
    double g1, g2, lam, lam2;
    if (NC) toon_gammas_nocld(tc, w0, g1, g2, lam, lam2);
    else toon_gammas(tc, w0, fcg, g1, g2, lam, lam2);
    const double E = fmin(lam * dt, clip);
    const double EP = fexp2(E * -NEG_LOG2E, K);
#if PZ_REFL_DIET
    // 1/g2 and 1/EP from one reciprocal (g2 of order w0 <= 1, EP <= e^35: the product stays in range; g2 = 0 -- a layer
    // that does not scatter -- is NaN in Gamma here as in round 4 and in the reference's (g1 - lamda)/g2, fluxes.py:1141)
    const double r_ge = frcp(g2 * EP);
    const double ig2 = r_ge * EP, EM = r_ge * g2;
    const double gam = (g1 - lam) * ig2;
#else
    const double gam = (g1 - lam) * frcp(g2);
    const double EM = frcp(EP);
#endif
    // No cloud anywhere in this layer of the wave (ftau_cld == 0 in every lane, checked per wave):
    // TTHG_ray's p_single = ftau_cld tthg + ftau_ray 3/4 (1 + cos^2) is its Rayleigh part alone
    // (bit-identical: 0 * tthg + x = x), without the two Henyey-Greenstein terms.
    double ps;
    if ((NC || L.nocld) && single_phase == 3)
        ps = L.fr * (0.75 * fma(cos_theta, cos_theta, 1.0));
    else
        ps = p_single<IS3D>(single_phase, L.cbo, L.gcos2, L.fc, L.fr, cos_theta, a.frac_a, a.frac_b,
                            frac_c, a.constant_back, a.constant_forward);
    const double ssa_h = (L.w0o * F * (0.125 / PI)) * ps;  // (w0_og F0PI/4pi) p_single / 2, fluxes.py:1397-1398
    const double w2pi = w0 * (0.5 / PI);                   // fluxes.py:1290-1296
    const double Fw0h = (0.5 * F) * w0;
    const double gcq = (multi_phase == 0) ? L.gcos2 : 0.0;   // N=2 vs N=1 (fluxes.py:1275-1287)
    // The direct-beam coefficients (fluxes.py:1146-1169) with g3 = 1/2 - c u0, g4 = 1/2 + c u0
    // (c = sqrt(3) fcg/2 quadrature, 3 fcg/4 Eddington) and u0 (1/u0) = 1 collapse to
    //   2 (g4 (g1 + 1/u0) + g2 g3) = A0 + hz,   2 (g3 (g1 - 1/u0) + g2 g4) = A0 - hz,
    //   A0 = g1 + g2 + 2c,  hz = 1/u0 + 2c (g1 - g2) u0,
    // and the multiple-scattering brackets (fluxes.py:1275-1296) with B0 = 1 + gcos2 q2 to
    //   (1 + 1.5 fcg u1 + q) + Gamma (1 - 1.5 fcg u1 + q) = (1 + Gamma) B0 + (1 - Gamma) 1.5 fcg u1  etc.
    const double c2 = NC ? 0.0 : ((tc == 1) ? 1.5 * fcg : SQ3 * fcg);
    const double A0 = NC ? (g1 + g2) : (g1 + g2) + c2;
    const double A1 = NC ? 0.0 : c2 * (g1 - g2);
    const double c15 = NC ? 0.0 : 1.5 * fcg;
    const double gp = 1.0 + gam;
    const double gmc = NC ? 0.0 : (1.0 - gam) * c15;
    const double w2gp = w2pi * gp, w2gmc = NC ? 0.0 : w2pi * gmc;      // PZ_REFL_DIET (2)
    const double w2A0 = w2pi * A0;                                    // PZ_REFL_DIET (3)

    // ---- elimination factors shared by all angles ----
    double a1i = 0.0, a2i = 0.0, ia = 0.0, rho_n = gam, sfac = 0.0;
    if (!FIRST) {
        const double em2 = S.pEM * S.pEM;
        const double a1 = fma(-(S.pgam * em2), S.rho, 1.0);
        const double a2 = fma(-em2, S.rho, S.pgam);
        const double d1 = fma(-gam, a2, a1);
        const double r12 = frcp(d1 * a1);                  // one reciprocal for 1/d1 and 1/a1
        const double inv = r12 * a1;
        a1i = a1 * inv;
        a2i = a2 * inv;
        rho_n = fma(gam, a1i, -a2i);
        ia = S.pEM * (r12 * d1);
        sfac = fma(-gam, rho_n, 1.0) * ia;
    }
    const double gEM = gam * EM;

    // One v_rcp_f64 (quarter rate) per angle for 1/den, 1/(lu-1) and 1/(lu+1): q_k = den_k (lu_k - 1)
    // (lu_k + 1).  Every factor is itself good to 1 ulp (lm1 is exact by Sterbenz near the lambda u1 = 1
    // singularity), so the results stay within a few ulp -- which matters: at lambda u1 -> 1 the
    // particular and homogeneous parts cancel with an amplification ~1/|lambda u1 - 1|.
    // (PZ_REFL_BATCH_RCP: all angles from one reciprocal of the running product, see the macro.)
    double den_[NA], lm1_[NA], lp1_[NA], lml_[NA], r3_[NA];
    {
        double q[NA], P[NA];
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            den_[k] = sub_unfused(lam2, g[k].iu0sq);   // lambda^2 - 1/u0^2, reference rounding
            const double lu = lam * g[k].u1;
            lm1_[k] = lu - 1.0;
            lp1_[k] = lu + 1.0;
            lml_[k] = lm1_[k] * lp1_[k];
            q[k] = den_[k] * lml_[k];
            P[k] = (k == 0) ? q[0] : P[k - 1] * q[k];
        }
#if PZ_REFL_BATCH_RCP
        double R = frcp(P[NA - 1]);
#pragma unroll
        for (int k = NA - 1; k > 0; --k) {
            r3_[k] = R * P[k - 1];
            R = R * q[k];
        }
        r3_[0] = R;
#else
#pragma unroll
        for (int k = 0; k < NA; ++k) r3_[k] = PZ_REFL_DIET ? frcp1(q[k]) : frcp(q[k]);
#endif
    }

#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const double u1_k = g[k].u1, u0_k = ZP ? g[k].u1 : g[k].u0, iu0_k = g[k].iu0;
        const double nl1_k = g[k].nl1, nl0_k = ZP ? g[k].nl1 : g[k].nl0, q2_k = g[k].q2;
        const double wq2_k = ZP ? 1.0 : g[k].wq2;
        const double den = den_[k], lm1 = lm1_[k], lp1 = lp1_[k], lml = lml_[k];
        const double r3 = r3_[k];
        const double rden = r3 * lml;                      // 1/den
        const double rd = r3 * den;                        // 1/((lu-1)(lu+1))
        const double hz = NC ? iu0_k : fma(A1, u0_k, iu0_k);
        const double am2 = A0 + hz, ap2 = A0 - hz;
        const double et = fexp2(dt * nl1_k, K);              // exp(-dtau/u1)
        const double e0 = ZP ? et : fexp2(dt * nl0_k, K);    // exp(-dtau/u0)
        const double xu = S.get(S_XU, k), Tk = S.get(S_T, k);
        const double xd = (AF || L.cum_tau) ? xu * e0 : fexp2_cold(L.tau_n * nl0_k, K);
        const double fw = Fw0h * rden;
        const double fx = fw * xu;
        const double cmu = am2 * fx, cpu = ap2 * fx;       // c-/c+ at the top of the layer
        double cmd, cpd;                                   // ... and at the bottom
        if (PZ_REFL_DIET && (AF || L.cum_tau)) {           // xd = xu e0: the bottom terms are the top ones times e0
            cmd = cmu * e0;
            cpd = cpu * e0;
        } else {
            const double fxd = fw * xd;
            cmd = am2 * fxd;
            cpd = ap2 * fxd;
        }
        S.set(S_XU, k, xd);
        // source-function coefficients (fluxes.py:1275-1296, 1395-1406)
        const double B0 = fma(gcq, q2_k, 1.0);
        const double h15 = NC ? 0.0 : c15 * u1_k;
        const double Aqq = NC ? B0 * A0 : fma(B0, A0, -(h15 * hz));       // (mpl c+ + mmi c-) / (2 fx)
        const double ee = fma(EP, et, -1.0), ff = fma(-EM, et, 1.0);
        double vp, vn;
        if (PZ_REFL_DIET) {
            // (x + 0 = x - 0 = x exactly: the cloud-free copy and the general body agree bit for bit on a layer without cloud)
            const double Trd = Tk * rd, wX = w2gp * B0;
            const double wY = NC ? 0.0 : w2gmc * u1_k;
            vp = ((Trd * (NC ? wX : wX + wY)) * lp1) * ee;
            vn = ((Trd * (NC ? wX : wX - wY)) * lm1) * ff;
        } else {
            const double X = gp * B0, Y = NC ? 0.0 : gmc * u1_k;
            const double Tw = Tk * w2pi, Trd = Tw * rd;
            vp = (Trd * lp1) * ((NC ? X : X + Y) * ee);
            vn = (Trd * lm1) * ((NC ? X : X - Y) * ff);
        }
        // exp(-tau_og[i]/u0): the running product of the layers above when tau_og really is the
        // running sum of dtau_og, else formed directly
        const double eo = (!FIRST && (AF || L.eo_ok)) ? S.get(S_EO, k) : fexp2_cold(L.tauo * nl0_k, K);
        // 1 - exp(-dtau (1/u0 + 1/u1)) and the same for dtau_og (fluxes.py:1395-1406) from the two
        // single-angle exponentials; a layer that is not delta-scaled (dtau_og == dtau) shares them
        const double t2 = fma(-e0, et, 1.0);
        double t1 = t2, e0o = e0;
        if (AF ? (SDT == 2) : !L.same_dt) {
            const double e1o = fexp2_cold(L.dto * nl1_k, K);
            e0o = ZP ? e1o : fexp2_cold(L.dto * nl0_k, K);
            t1 = fma(-e0o, e1o, 1.0);
        }
        if (!LAST) S.set(S_EO, k, eo * e0o);
        // S0 = (ssa eo t1 + Aq t2) u0/(u0+u1) with Aq = 2 w2pi fx Aqq; wq2 = 2 u0/(u0+u1) (1 if ZP)
        double S0;
        if (PZ_REFL_DIET && ZP && (NC || L.nocld) && (AF ? (SDT != 2) : L.same_dt)) {
            // t1 == t2 and Aqq = B0 A0 (no cloud: the general body's h15 hz is +0): S0 = t2 (ssa eo + (w0/2pi A0) (B0 fx))
            S0 = fma(w2A0, B0 * fx, ssa_h * eo) * t2;
        } else {
            const double s1 = (ssa_h * wq2_k) * (eo * t1);
            S0 = fma((w2pi * wq2_k) * fx, Aqq * t2, s1);
        }
        double kap = fma(Tk, S0, S.get(S_KAPPA, k));
        const double Tn = Tk * et;
        if (LAST) {                                        // xint[n] = flux_zero/pi (fluxes.py:1266-1270)
            vp = fma(Tn * EP, 1.0 / PI, vp);
            vn = fma((Tn * gam) * EM, 1.0 / PI, vn);
            kap = fma(Tn * cpd, 1.0 / PI, kap);
        }
        double delta_n;
        if (FIRST) {                                       // top row (fluxes.py:155-158)
            delta_n = b_top - cmu;
            S.set(S_ZETA, k, fma(-vn, gam, vp));
            kap = fma(vn, delta_n, kap);
        } else {
            const double rP = cpu - S.get(S_D1, k);
            const double rM = cmu - S.get(S_D2, k);
            const double zeta = S.get(S_ZETA, k);
            delta_n = fma(a2i, rP, -(a1i * rM));
            const double t = fma(gam, delta_n, rP) * ia;
            kap = fma(zeta, t, fma(vn, delta_n, kap));
            S.set(S_ZETA, k, fma(-vn, rho_n, fma(zeta, sfac, vp)));
        }
        S.set(S_KAPPA, k, kap);
        S.set(S_T, k, Tn);
        S.set(S_D1, k, fma(gEM, delta_n, cpd));
        S.set(S_D2, k, fma(EM, delta_n, cmd));
    }
    S.rho = rho_n;
    S.pgam = gam;
    S.pEM = EM;
}

// One or two angles per lane (the 3-D facet kernel, angle tails) need far fewer registers: ask for
// more waves per SIMD there, the HBM-bound case lives on memory-level parallelism.
#ifndef PZ_REFL_MINWAVES_FEW
#define PZ_REFL_MINWAVES_FEW 2
#endif
// albedo + xint*gweight*tweight in the reference's operation order, unfused (disco.py:145-146), so the
// fused disk sum, k_compress and numpy agree bit for bit
__device__ __forceinline__ double disk_accumulate(double acc, double x, double gw, double tw)
{
#pragma clang fp contract(off)
    return acc + x * gw * tw;
}

// sym_fac*0.5*albedo/F0PI*(cos_theta+1) (disco.py:148), the same operations as k_compress
__device__ __forceinline__ double disk_finish(double c1, double acc, double F, double cos_theta)
{
#pragma clang fp contract(off)
    return c1 * acc / F * (cos_theta + 1.0);
}

// BIG: one wave per SIMD (grids of up to 1 024 column-waves, five angles): the whole 512-entry register file
// belongs to the wave, so the sweep state stays in registers instead of LDS -- 50 000 columns 0.1365 ms against
// 0.1636 (round-2 sweep), same bits.  Of no use to the 1e5-column launch: its 1 563 waves need
// two per SIMD.
// DRV (3-D only): some planes are NULL and re-derived in the kernel (see below); the full-plane 3-D launch keeps
// the branch-free load sequence (the conditional loads cost the HBM-bound facet kernel 7 %)
// The kernel body.  `bx_in` = this workgroup's column group within its spectrum, `by_in` = its angle group,
// `angp` = the angle table (the kernel argument's, or the batch entry's).
template <int NA, bool IS3D, bool ZP, bool FAST, bool BIG, int DRV, typename AnglePtr>
__device__ __forceinline__ void reflected_toa_body(const ReflectedArgs &a, const unsigned bx_in, const unsigned by_in,
                                                   AnglePtr angp)
{
#pragma clang fp contract(off)      // operations as written, see reflected_layer
    const unsigned bx = bx_in, by = by_in;
    const long col = bx * (long)blockDim.x + threadIdx.x;
    if (col >= a.ncol) return;
    const int nfac = IS3D ? a.nfac : 1;
    const long w = IS3D ? col / nfac : (a.ncolper > 1 ? col / a.ncolper : col);
    const int fac = IS3D ? (int)(col - w * nfac) : 0;
    const int n = a.nlayer;
    const long pitch = a.pitch;
    const double clip = IS3D ? 40.0 : 35.0;            // fluxes.py:516 vs :1174
    const int tc = IS3D ? 0 : a.toon_coefficients;     // 3-D is quadrature only (fluxes.py:489)
    const double b_top = IS3D ? 0.0 : a.b_top;         // fluxes.py:522

    Exp2Coef K;
    K.load();
    ReflectedArgs::Angle g[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        if (IS3D) {
            const double v0 = fabs(a.u0_tab[fac]), v1 = fabs(a.u1_tab[fac]);   // fluxes.py:467-468
            g[k].u0 = v0;
            g[k].u1 = v1;
            g[k].iu0 = 1.0 / v0;
            g[k].iu0sq = 1.0 / (v0 * v0);              // as the reference forms it (fluxes.py:1155)
            g[k].nl0 = NEG_LOG2E / v0;
            g[k].nl1 = NEG_LOG2E / v1;
            g[k].nlm = NEG_LOG2E * (1.0 / v0 + 1.0 / v1);
            g[k].wq2 = 2.0 * (v0 / (v0 + v1));
            const double ubar2 = 0.767;                 // fluxes.py:1280
            g[k].q2 = (3.0 * ubar2 * ubar2 * v1 * v1 - 1.0) / 2.0;
            g[k].wgt = g[k].wgt2 = 0.0;
        } else {                                        // host-precomputed, wave-uniform
            const auto &src = angp[by * NA + k];        // by: angle group (0 unless ny > 1)
            g[k].u1 = src.u1; g[k].iu0 = src.iu0; g[k].iu0sq = src.iu0sq; g[k].nl1 = src.nl1; g[k].q2 = src.q2;
            g[k].u0 = src.u0; g[k].nl0 = src.nl0; g[k].nlm = src.nlm; g[k].wq2 = src.wq2; g[k].wgt = src.wgt;
            g[k].wgt2 = src.wgt2;
        }
    }
    const double F = a.F0PI[w], rs = a.surf_reflect[w];

    const double *p_dtau = a.dtau + col, *p_tau = a.tau + col, *p_w0 = a.w0 + col,
                 *p_cosb = a.cosb + col, *p_gcos2 = a.gcos2 + col, *p_fc = a.ftau_cld + col,
                 *p_fr = a.ftau_ray + col, *p_dto = a.dtau_og + col, *p_tauo = a.tau_og + col,
                 *p_w0o = a.w0_og + col, *p_cbo = a.cosb_og + col;

#ifndef PZ_REFL_LDS_MIN
#define PZ_REFL_LDS_MIN 4
#endif
    constexpr bool LDS = (NA >= PZ_REFL_LDS_MIN) && !IS3D && !BIG;
    __shared__ double lds_state[LDS ? (PZ_REFL_NLDS > 0 ? PZ_REFL_NLDS : 1) * NA * PZ_REFL_BLOCK : 1];
    ReflState<NA, LDS> S;
    S.lds = lds_state + threadIdx.x;
    S.rho = S.pgam = S.pEM = 0.0;
    // 3-D entry point: planes that compute_opacity derives exactly from others may be left out (NULL) and
    // are re-derived here with the same operations instead of being written to and read from HBM --
    //   tau / tau_og   : running sums of dtau / dtau_og from 0 at the top (optics.py:353-354, 418-420)
    //   gcos2          : 0.5 ftau_ray (optics.py:342)
    //   ftau_cld (with cosb, cosb_og, ftau_ray, gcos2): column without cloud: 0, 0, 0, 1, 0.5 (optics.py:335-342
    //                    with TAUCLD = 0)
    //   dtau_og / w0_og: no delta-scaling (cosb = 0: f = 0, optics.py:412-420 reduce to x*1): dtau / w0
    // all wave-uniform (kernel arguments).  The 1-D default-options launches take the same patterns (round 4: the product
    // leaves tau / tau_og / gcos2 -- and for a cloud-free atmosphere everything but dtau and w0 -- out of HBM; the caller
    // asks picaso_reflected_1d_can_derive first); every other 1-D launch is handed all eleven planes
    // DRV = 2 (1-D default options): the launcher saw that ONLY dtau and w0 exist -- the pattern of an atmosphere without
    // cloud -- and says so at compile time: no flag, no test and no second copy of the layer body survive in this variant
    // DRV = 3: the other pattern of the product (a cloudy atmosphere): tau, tau_og and gcos2 left out, the cloud planes
    // present -- known at compile time, so the two tests of "the level depths are the running sums" (they are: they are
    // formed here), the exec-mask tests and the branches behind them disappear from every layer (timing build of the
    // eleven-plane kernel with those two tests compiled out: 0.223 -> 0.207 ms)
    constexpr bool CLEAR = (DRV == 2), LEVELS = (DRV == 3);
    const bool derive_tau = CLEAR || LEVELS || (DRV && a.tau == nullptr);
    const bool derive_tauo = CLEAR || LEVELS || (DRV && a.tau_og == nullptr);
    const bool derive_g2 = CLEAR || LEVELS || (DRV && a.gcos2 == nullptr);
    const bool clear = CLEAR || (!LEVELS && DRV && a.ftau_cld == nullptr);
    const bool alias_og = CLEAR || (!LEVELS && DRV && a.dtau_og == nullptr);
    double tau_i = derive_tau ? 0.0 : p_tau[0];
    double tauo_pred = 0.0;          // tau_og[i-1] + dtau_og[i-1] of the layer above
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        S.set(S_T, k, 1.0);
        S.set(S_KAPPA, k, 0.0);
        S.set(S_ZETA, k, 0.0);
        S.set(S_D1, k, 0.0);
        S.set(S_D2, k, 0.0);
        S.set(S_EO, k, 0.0);
        S.set(S_XU, k, fexp2(mul_unfused(tau_i, ZP ? g[k].nl1 : g[k].nl0), K));
    }

    // Software prefetch with two register sets used alternately (loop unrolled by two): the eleven
    // plane loads of layer i+1 are issued before layer i is computed and nothing has to be copied
    // between the sets, so they stay in flight for a whole layer (a copy would force the wave to
    // drain vmcnt at the end of every layer, which costs the HBM-bound facet kernel ~25 %).
    // FAST: planes addressed as (uniform base) + (32-bit lane offset): one v_add_u32 per layer for all
    // eleven loads (global_load ... saddr) instead of eleven 64-bit lane address adds (-2.5 % time)
    const unsigned voff0 = (unsigned)(col * 8);
    const unsigned pitch8 = (unsigned)(pitch * 8);
    auto ld = [&](const double *base, unsigned off) {
        return *(const double *)((const char *)base + off);
    };
    auto load = [&](LayerIn &L, int i) {
        if constexpr (FAST && PZ_REFL_SADDR && DRV) {      // 1-D with planes left out: only what exists is loaded
            const unsigned o = voff0 + (unsigned)i * pitch8;
            L.dt = ld(a.dtau, o);
            L.w0 = ld(a.w0, o);
            if constexpr (CLEAR) return;
            if constexpr (LEVELS) {
                L.g = ld(a.cosb, o);
                L.fc = ld(a.ftau_cld, o);
                L.fr = ld(a.ftau_ray, o);
                L.cbo = ld(a.cosb_og, o);
                L.dto = ld(a.dtau_og, o);
                L.w0o = ld(a.w0_og, o);
                return;
            }
            if (!derive_tau) L.tau_n = ld(a.tau + pitch, o);
            if (!clear) {
                L.g = ld(a.cosb, o);
                L.fc = ld(a.ftau_cld, o);
                L.fr = ld(a.ftau_ray, o);
                L.cbo = ld(a.cosb_og, o);
                if (!derive_g2) L.gcos2 = ld(a.gcos2, o);
            }
            if (!alias_og) {
                L.dto = ld(a.dtau_og, o);
                L.w0o = ld(a.w0_og, o);
            }
            if (!derive_tauo) L.tauo = ld(a.tau_og, o);
            return;
        }
        if constexpr (FAST && PZ_REFL_SADDR) {
            const unsigned o = voff0 + (unsigned)i * pitch8;
            L.dt = ld(a.dtau, o);
            L.tau_n = ld(a.tau + pitch, o);
            L.w0 = ld(a.w0, o);
            L.g = ld(a.cosb, o);
            L.gcos2 = ld(a.gcos2, o);
            L.fc = ld(a.ftau_cld, o);
            L.fr = ld(a.ftau_ray, o);
            L.dto = ld(a.dtau_og, o);
            L.tauo = ld(a.tau_og, o);
            L.w0o = ld(a.w0_og, o);
            L.cbo = ld(a.cosb_og, o);
            return;
        }
        const long o = (long)i * pitch;
        L.dt = p_dtau[o];
        L.w0 = p_w0[o];
        if constexpr (DRV) {                       // only the loads are issued here; derived values in prep
            if (!derive_tau) L.tau_n = p_tau[o + pitch];
            if (!clear) {
                L.g = p_cosb[o];
                L.fc = p_fc[o];
                L.fr = p_fr[o];
                L.cbo = p_cbo[o];
                if (!derive_g2) L.gcos2 = p_gcos2[o];
            }
            if (!alias_og) {
                L.dto = p_dto[o];
                L.w0o = p_w0o[o];
            }
            if (!derive_tauo) L.tauo = p_tauo[o];
            return;
        }
        L.tau_n = p_tau[o + pitch];
        L.g = p_cosb[o];
        L.gcos2 = p_gcos2[o];
        L.fc = p_fc[o];
        L.fr = p_fr[o];
        L.dto = p_dto[o];
        L.tauo = p_tauo[o];
        L.w0o = p_w0o[o];
        L.cbo = p_cbo[o];
    };
    auto prep = [&](LayerIn &L) {
        if constexpr (CLEAR) {
            L.g = 0.0; L.fc = 0.0; L.fr = 1.0; L.gcos2 = 0.5; L.cbo = 0.0;
            L.dto = L.dt; L.w0o = L.w0;
            L.tau_n = tau_i + L.dt;
            L.tauo = tauo_pred;
            L.cum_tau = L.eo_ok = L.same_dt = L.nocld = L.allf = true;
            L.af_cloud = false;
            tau_i = L.tau_n;
            tauo_pred = L.tauo + L.dto;
            return;
        }
        if constexpr (LEVELS) {
            L.gcos2 = 0.5 * L.fr;
            L.tau_n = tau_i + L.dt;
            L.tauo = tauo_pred;
            L.cum_tau = L.eo_ok = true;
            L.same_dt = __all(L.dto == L.dt);
            L.nocld = __all(L.fc == 0.0);
            L.allf = L.same_dt;
            L.af_cloud = !L.same_dt && !L.nocld;
            tau_i = L.tau_n;
            tauo_pred = L.tauo + L.dto;
            return;
        }
        if constexpr (DRV) {
            if (clear) { L.g = 0.0; L.fc = 0.0; L.fr = 1.0; L.gcos2 = 0.5; L.cbo = 0.0; }
            else if (derive_g2) L.gcos2 = 0.5 * L.fr;
            if (alias_og) { L.dto = L.dt; L.w0o = L.w0; }
            if (derive_tau) L.tau_n = tau_i + L.dt;
            if (derive_tauo) L.tauo = tauo_pred;
        }
        L.cum_tau = __all(L.tau_n == tau_i + L.dt);
        L.eo_ok = __all(L.tauo == tauo_pred);
        L.same_dt = __all(L.dto == L.dt);
        L.nocld = __all(L.fc == 0.0);
        L.allf = L.cum_tau && L.eo_ok && L.same_dt;
        L.af_cloud = L.cum_tau && L.eo_ok && !L.same_dt && !L.nocld;
        tau_i = L.tau_n;
        tauo_pred = L.tauo + L.dto;
    };
#define PZ_LAYER(FIRST_, LAST_, L_)                                                                      \
    do {                                                                                                 \
        prep(L_);                                                                                        \
        if (PZ_REFL_NOCLD_BODY && FAST && !(FIRST_) && !(LAST_) && L_.nocld && L_.allf)                  \
            reflected_layer<NA, IS3D, ZP, FIRST_, LAST_, LDS, FAST, true, true>(a, L_, S, g, K, F, clip, tc, b_top); \
        else if (PZ_REFL_CLOUD_BODY && FAST && !(FIRST_) && !(LAST_) && L_.af_cloud)                      \
            reflected_layer<NA, IS3D, ZP, FIRST_, LAST_, LDS, FAST, false, true, 2>(a, L_, S, g, K, F, clip, tc, b_top); \
        else                                                                                             \
            reflected_layer<NA, IS3D, ZP, FIRST_, LAST_, LDS, FAST>(a, L_, S, g, K, F, clip, tc, b_top); \
        if constexpr (CLEAR) __builtin_amdgcn_sched_barrier(0);   /* straight-line code: keep the layers apart */ \
    } while (0)

    // Two register sets used alternately (no copy) for one or two angles per lane -- the HBM-bound 3-D facet
    // kernel -- and for five: there the eleven v_mov_b64 of the copy are 2.3 % of the launch (0.2357 ->
    // 0.2302 ms, same box, no spills at 254 VGPRs).  Three angles spill with a second set (14 VGPRs), four
    // and six to eight (chunked launches, rarely used) are left on the copying form.  (Early in round 1 the
    // second set made the five-angle kernel spill: 0.98 ms vs 0.36 ms; the state has since moved to LDS.)
#ifndef PZ_REFL_TWOSETS_5
#define PZ_REFL_TWOSETS_5 1
#endif
#ifndef PZ_REFL_TWOSETS_NONZP
#define PZ_REFL_TWOSETS_NONZP 0
#endif
    constexpr bool TWO_SETS = (NA <= 2) || (PZ_REFL_TWOSETS_5 && NA == 5 && FAST && (ZP || PZ_REFL_TWOSETS_NONZP));   // generic five-angle kernel: 30 VGPRs spilled with it
    LayerIn A, B;
    load(A, 0);
    if (n == 1) {
        PZ_LAYER(true, true, A);
    } else if constexpr (TWO_SETS) {
        load(B, 1);
        PZ_LAYER(true, false, A);
        int i = 1;                               // B holds layer i
        for (; i + 2 <= n - 1; i += 2) {         // layers i and i+1 are interior, layer i+2 exists
            load(A, i + 1);
            PZ_LAYER(false, false, B);
            load(B, i + 2);
            PZ_LAYER(false, false, A);
        }
        if (i == n - 1) {
            PZ_LAYER(false, true, B);
        } else {                                 // i == n - 2
            load(A, i + 1);
            PZ_LAYER(false, false, B);
            PZ_LAYER(false, true, A);
        }
    } else {
        load(B, 1);                              // B: prefetch set, A: the layer being computed
        PZ_LAYER(true, false, A);
        for (int i = 1; i < n - 1; ++i) {
            A = B;
            load(B, i + 1);
            PZ_LAYER(false, false, A);
        }
        A = B;
        PZ_LAYER(false, true, A);
    }
#undef PZ_LAYER

    // ---- surface row (fluxes.py:178-183) and output ----
    // EP(1 - rs G) pos + EM(G - rs) neg = b_surface - c+dn + rs c-dn with neg = delta - rho pos,
    // divided through by EP:  pos = EM (b_surface - D1 + rs D2) / ((1 - rs G) - EM^2 (G - rs) rho)
    // disk sum in the reference's order (one running sum over all angles): a later chunk of a launch
    // split over angles continues from the stored partial sum
    double alb = (!IS3D && a.albedo && !a.albedo_first) ? a.albedo[w] : 0.0;
    double xout[NA];
    {
#pragma clang fp contract(off)
        const double em2 = S.pEM * S.pEM;
        const double bden = 1.0 / fma(-(em2 * (S.pgam - rs)), S.rho, fma(-rs, S.pgam, 1.0));
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const double b_surface = ((rs * (ZP ? g[k].u1 : g[k].u0)) * F) * S.get(S_XU, k);
            const double pos = (S.pEM * fma(rs, S.get(S_D2, k), b_surface - S.get(S_D1, k))) * bden;
            xout[k] = fma(S.get(S_ZETA, k), pos, S.get(S_KAPPA, k));
        }
    }
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const double x = xout[k];
        if (IS3D) a.xint[(long)fac * a.nwno + w] = x;
        else if (NA == 1 || (int)by * NA + k < a.nvalid) a.xint[(long)(by * NA + k) * a.ncol + col] = x;
        alb = disk_accumulate(alb, x, g[k].wgt, g[k].wgt2);
    }
    if (!IS3D && a.albedo) {                              // fused disco.compress_disco (disco.py:145-148)
        double acc = alb;
        if (a.albedo_last) acc = disk_finish(a.albedo_scale, acc, F, a.cos_theta);
        a.albedo[w] = acc;
    }
}

// Workgroup b of a 1-D grid of `nrep` replicas (angle groups, or spectra that share their planes) of `ncg` column
// groups -> (replica, column group) in XCD-aware order, see the comment in k_reflected_toa.
__device__ __forceinline__ void xcd_decode(unsigned b, unsigned nrep, unsigned ncg, unsigned &rep, unsigned &cg)
{
    const unsigned nfull = ncg & ~7u;
    if (b < nfull * nrep) {
        const unsigned per = 8u * nrep, chunk = b / per, rem = b - chunk * per;
        rep = rem >> 3;
        cg = chunk * 8u + (rem & 7u);
    } else {
        const unsigned idx = b - nfull * nrep, left = ncg - nfull;
        rep = idx / left;
        cg = nfull + (idx - rep * left);
    }
}

template <int NA, bool IS3D, bool ZP, bool FAST = false, bool BIG = false, int DRV = 0>
__global__ __launch_bounds__(PZ_REFL_BLOCK, (BIG ? 1 : NA <= 2 ? PZ_REFL_MINWAVES_FEW : PZ_REFL_MINWAVES)) void k_reflected_toa(const ReflectedArgs a)
{
    // Angle groups as separate workgroups (ny > 1): every group re-reads the eleven planes of its columns.
    // Consecutive workgroups go to consecutive XCDs (8, each with its own L2), so the launch is 1-D and, for
    // the column groups that fill whole chunks of 8, block b = (chunk, angle group, column group within the
    // chunk): the ny workgroups of one column group are dispatched 8 apart, all to XCD (column group % 8),
    // and all but the first find the planes in that L2.  The last ncg % 8 column groups go out in plain
    // (angle group, column group) order, which spreads their workgroups over the XCDs -- sending them after
    // their chunk would put up to 8 x ny more workgroups on the first XCDs than on the others (12 500
    // columns, one angle per wave: 35 on XCD 0 of 32 CUs, 0.078 ms instead of 0.050).  A 2-D (x, y) grid
    // shares L2 only when the column-group count happens to be a multiple of 8 (12 250 columns 0.050 ms,
    // 12 000 0.069, 12 500 0.072: round-2 sweep).
    unsigned bx = blockIdx.x, by = 0;
    if (!IS3D && a.ny > 1) xcd_decode(blockIdx.x, (unsigned)a.ny, (unsigned)a.ncg, by, bx);
    reflected_toa_body<NA, IS3D, ZP, FAST, BIG, DRV>(a, bx, by, a.ang);
}

// The same body for `nspec` spectra in one grid (picaso_get_reflected_1d_batch_dev): the workgroup finds its
// spectrum's planes, outputs and angle table in the device table a.batch.  A separate instantiation so that the
// single-spectrum kernels keep their register allocation; same operations, same bits.
template <int NA, bool IS3D, bool ZP, bool FAST = false, bool BIG = false, int DRV = 0>
__global__ __launch_bounds__(PZ_REFL_BLOCK, (BIG ? 1 : NA <= 2 ? PZ_REFL_MINWAVES_FEW : PZ_REFL_MINWAVES)) void k_reflected_toa_batch(const ReflectedArgs a)
{
    unsigned spec, bx, by = 0;
    const unsigned ny = a.ny > 1 ? (unsigned)a.ny : 1u;
    if (!IS3D && a.batch_interleave) {          // shared planes: (spectrum, angle group) replicas of a column group on one XCD
        unsigned rep;
        xcd_decode(blockIdx.x, ny * (unsigned)a.nspec, (unsigned)a.ncg, rep, bx);
        spec = rep / ny;
        by = rep - spec * ny;
    } else {
        spec = blockIdx.x / a.bps;
        bx = blockIdx.x - spec * a.bps;
        if (!IS3D && ny > 1) xcd_decode(bx, ny, (unsigned)a.ncg, by, bx);
    }
    // The table is read through the constant address space (it is written before the launch and never by it):
    // its entries then behave like kernel arguments -- scalar loads the compiler may repeat instead of keeping
    // ~60 angle constants alive through the layer loop (as plain global loads they ended up in VGPRs: 98 spilled).
    typedef const __attribute__((address_space(4))) ReflBatchItem *ItemPtr;
    const auto &it = *((ItemPtr)(unsigned long)a.batch + spec);
    ReflectedArgs b = a;
    b.dtau = it.dtau; b.tau = it.tau; b.w0 = it.w0; b.cosb = it.cosb; b.gcos2 = it.gcos2; b.ftau_cld = it.ftau_cld;
    b.ftau_ray = it.ftau_ray; b.dtau_og = it.dtau_og; b.tau_og = it.tau_og; b.w0_og = it.w0_og; b.cosb_og = it.cosb_og;
    b.surf_reflect = it.surf_reflect; b.F0PI = it.F0PI; b.xint = it.xint; b.albedo = it.albedo;
    b.cos_theta = it.cos_theta;
    b.u0_tab = it.u0_tab; b.u1_tab = it.u1_tab;
    reflected_toa_body<NA, IS3D, ZP, FAST, BIG, DRV>(b, bx, by, it.ang);
}

// the reference's default options (see reflected_layer); zp: the symmetric zero-phase geometry, where the
// compile-time variant also fixes cos_theta = 1 -- at other phase angles (ubar0 != ubar1) cos_theta stays an argument
static bool fast_options(const ReflectedArgs &a, bool zp)
{
    if (getenv("PICASO_AMD_REFL_GENERIC")) return false;      // A/B switch for tools/
    if ((double)a.pitch * (a.nlayer + 1) * 8.0 >= 4294967296.0) return false;   // 32-bit plane offsets
    // (a batched launch: a.cos_theta = 1 only if every spectrum's is)
    return a.toon_coefficients == 0 && a.single_phase == 3 && a.multi_phase == 0 && (!zp || a.cos_theta == 1.0) &&
           a.frac_c == 2.0;
}

template <int NA>
static int launch1d(picaso_ctx *ctx, const ReflectedArgs &a_in)
{
    ReflectedArgs a = a_in;
    if (a.nvalid <= 0) a.nvalid = NA * (a.ny > 1 ? a.ny : 1);     // no padded angle slots
    const int block = PZ_REFL_BLOCK;
    const unsigned ncg = (unsigned)((a.ncol + block - 1) / block), ny = (unsigned)(a.ny > 1 ? a.ny : 1);
    const unsigned nspec = a.batch ? (unsigned)a.nspec : 1u;
    a.ncg = (int)ncg;
    a.bps = ncg * ny;
    const dim3 grid(ncg * ny * nspec);              // ny > 1: 1-D, in the XCD-aware order of the kernel
    bool zp = a.batch ? a.batch_zp != 0 : true;     // batch: the angle tables are on the device; the caller looked
    if (!a.batch)
        for (int k = 0; k < a.na * (int)ny; ++k) zp = zp && (a.ang[k].u0 == a.ang[k].u1);
    bool big = false;
    if constexpr (NA == 5)
        big = a.ny <= 1 && (long)nspec * ncg * (block / 64) <= 1024 && getenv("PICASO_AMD_REFL_NO_BIG") == nullptr;
    bool fast = fast_options(a, zp);
#define PZ_GO(KERNEL) hipLaunchKernelGGL(KERNEL, grid, dim3(block), 0, ctx->stream, a)
    // planes left out (re-derived in the kernel): the default-options kernels only -- reflected_1d_can_derive() is the
    // caller's way to know
    const bool drv = !(a.tau && a.tau_og && a.gcos2 && a.ftau_cld && a.dtau_og);
    if (drv) {
        // reflected_1d_can_derive() looked at ALL angles of the call; this launch may be a chunk of them whose angles
        // all happen to have ubar0 == ubar1 at a phase angle other than zero: not the zero-phase variant's geometry
        // (it fixes cos_theta = 1), but the general default-options kernel takes it -- same bits
        if (!fast && zp && a.cos_theta != 1.0) {
            zp = false;
            fast = fast_options(a, zp);
        }
        if (!fast) return fail(ctx, "reflected: planes may be left out only with the reference's default options");
        // only dtau and w0: the compile-time form (five angles, symmetric geometry: the shape of a spectrum() call)
        const bool only2 = !a.tau && !a.tau_og && !a.gcos2 && !a.ftau_cld && !a.dtau_og && PZ_REFL_CLEAR_VARIANT;
        const bool levels3 = !a.tau && !a.tau_og && !a.gcos2 && a.ftau_cld && a.dtau_og && PZ_REFL_LEVELS_VARIANT;
        if (a.batch) {
            if (zp && big) { if constexpr (NA == 5) PZ_GO((k_reflected_toa_batch<NA, false, true, true, true, 1>)); }
            else if (zp && only2 && NA == 5) { if constexpr (NA == 5) PZ_GO((k_reflected_toa_batch<NA, false, true, true, false, 2>)); }
            else if (zp && levels3 && NA == 5) { if constexpr (NA == 5) PZ_GO((k_reflected_toa_batch<NA, false, true, true, false, 3>)); }
            else if (zp) PZ_GO((k_reflected_toa_batch<NA, false, true, true, false, 1>));
            else PZ_GO((k_reflected_toa_batch<NA, false, false, true, false, 1>));
        } else if (zp && big) { if constexpr (NA == 5) PZ_GO((k_reflected_toa<NA, false, true, true, true, 1>)); }
        else if (zp && only2 && NA == 5) { if constexpr (NA == 5) PZ_GO((k_reflected_toa<NA, false, true, true, false, 2>)); }
        else if (zp && levels3 && NA == 5) { if constexpr (NA == 5) PZ_GO((k_reflected_toa<NA, false, true, true, false, 3>)); }
        else if (zp) PZ_GO((k_reflected_toa<NA, false, true, true, false, 1>));
        else PZ_GO((k_reflected_toa<NA, false, false, true, false, 1>));
        PZ_HIP(ctx, hipGetLastError());
        return 0;
    }
    if (a.batch) {
        if (zp && fast && big) {
            if constexpr (NA == 5) PZ_GO((k_reflected_toa_batch<NA, false, true, true, true>));
        } else if (zp && fast) PZ_GO((k_reflected_toa_batch<NA, false, true, true>));
        else if (fast && PZ_REFL_FAST_NONZP) PZ_GO((k_reflected_toa_batch<NA, false, false, true>));
        else if (zp) PZ_GO((k_reflected_toa_batch<NA, false, true>));
        else PZ_GO((k_reflected_toa_batch<NA, false, false>));
    } else if (zp && fast && big) {
        if constexpr (NA == 5) PZ_GO((k_reflected_toa<NA, false, true, true, true>));
    } else if (zp && fast)
        PZ_GO((k_reflected_toa<NA, false, true, true>));
    else if (fast && PZ_REFL_FAST_NONZP)        // default options at a non-zero phase angle
        PZ_GO((k_reflected_toa<NA, false, false, true>));
    else if (zp)
        PZ_GO((k_reflected_toa<NA, false, true>));
    else
        PZ_GO((k_reflected_toa<NA, false, false>));
#undef PZ_GO
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

int launch_reflected_toa(picaso_ctx *ctx, const ReflectedArgs &a, bool is3d)
{
    if (a.ncol <= 0 || a.nlayer < 1) return fail(ctx, "reflected: empty problem");
    // 3-D: the reference's default options (TTHG_ray, N = 2, frac_c = 2; 3-D is quadrature-only anyway) fixed at compile
    // time like the 1-D launches, and with them the two plane patterns of the product known at compile time as well --
    // only dtau and w0 (a map without cloud), or everything but tau / tau_og / gcos2 (cloud tables): same bits as the
    // generic instantiations (PICASO_AMD_REFL3D_GENERIC=1), which every other option set and plane pattern keeps
    const bool all3 = a.tau && a.tau_og && a.gcos2 && a.ftau_cld && a.dtau_og;
    const bool fast3 = is3d && a.single_phase == 3 && a.multi_phase == 0 && a.frac_c == 2.0 &&
                       (double)a.pitch * (a.nlayer + 1) * 8.0 < 4294967296.0 && !getenv("PICASO_AMD_REFL3D_GENERIC");
    const bool only2_3 = !a.tau && !a.tau_og && !a.gcos2 && !a.ftau_cld && !a.dtau_og;
    const bool levels3_3 = !a.tau && !a.tau_og && !a.gcos2 && a.ftau_cld && a.dtau_og;
    if (is3d && a.batch) {                         // a.tau etc.: the presence pattern of EVERY spectrum's planes
        const int block = PZ_REFL_BLOCK;
        ReflectedArgs b = a;
        b.bps = (unsigned)((a.ncol + block - 1) / block);
        b.batch_interleave = 0;
        const dim3 grid(b.bps * (unsigned)a.nspec);
#define PZ_GO3(KERNEL) hipLaunchKernelGGL(KERNEL, grid, dim3(block), 0, ctx->stream, b)
        if (fast3 && all3) PZ_GO3((k_reflected_toa_batch<1, true, false, true, false, 0>));
        else if (fast3 && only2_3) PZ_GO3((k_reflected_toa_batch<1, true, false, true, false, 2>));
        else if (fast3 && levels3_3) PZ_GO3((k_reflected_toa_batch<1, true, false, true, false, 3>));
        else if (fast3) PZ_GO3((k_reflected_toa_batch<1, true, false, true, false, 1>));
        else if (all3) PZ_GO3((k_reflected_toa_batch<1, true, false>));
        else PZ_GO3((k_reflected_toa_batch<1, true, false, false, false, 1>));
#undef PZ_GO3
        PZ_HIP(ctx, hipGetLastError());
        return 0;
    }
    if (is3d) {
        const int block = PZ_REFL_BLOCK;
        const dim3 grid((unsigned)((a.ncol + block - 1) / block));
#define PZ_GO3(KERNEL) hipLaunchKernelGGL(KERNEL, grid, dim3(block), 0, ctx->stream, a)
        if (fast3 && all3) PZ_GO3((k_reflected_toa<1, true, false, true, false, 0>));
        else if (fast3 && only2_3) PZ_GO3((k_reflected_toa<1, true, false, true, false, 2>));
        else if (fast3 && levels3_3) PZ_GO3((k_reflected_toa<1, true, false, true, false, 3>));
        else if (fast3) PZ_GO3((k_reflected_toa<1, true, false, true, false, 1>));
        else if (all3) PZ_GO3((k_reflected_toa<1, true, false>));
        else PZ_GO3((k_reflected_toa<1, true, false, false, false, 1>));
#undef PZ_GO3
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
    return fail(ctx, "reflected: unsupported angle chunk %d", a.na);
}

}  // namespace pz
