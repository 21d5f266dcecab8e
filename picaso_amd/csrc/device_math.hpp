// Device-side helpers shared by the Toon kernels (gfx950, fp64).
#pragma once
#include <hip/hip_runtime.h>

namespace pz {

constexpr double PI = 3.14159265358979323846;
constexpr double SQ3 = 1.7320508075688772935;

// ---------------------------------------------------------------------------------------------
// Lean fp64 transcendental helpers.  The kernels are FP64-VALU bound (about 20 fp64 instructions
// per algorithmic byte at 5 angles), so the instruction count of exp and divide sets the speed.
// ---------------------------------------------------------------------------------------------

// exp(x), <= 2 ulp, 17 VALU instructions (ocml's exp is ~25 with its overflow/underflow selects).
// Range reduction x = k ln2 + r with a two-term ln2, degree-11 polynomial fitted on
// |r| <= ln2/2 (Chebyshev-node interpolant of (e^r - 1 - r)/r^2, max rel. error 2.2e-16 incl.
// evaluation rounding), scaling by v_ldexp_f64 which also gives the correct gradual underflow
// to 0 for very negative x and +inf for x > 709.8.  NaN propagates.  (exp(-inf) gives NaN, not
// 0: the solver never forms it for finite optical depths.)
// p*r + c as a 3-address VOP3 v_fma_f64 with the coefficient in an SGPR pair.  Written as inline
// asm because hipcc otherwise selects the 2-address v_fmac_f64 for Horner steps and has to re-copy
// the (VGPR-parked) coefficient into the destination before every step: one v_mov_b64 per
// polynomial term, ~230 dead moves per layer in the 5-angle reflected kernel.
// Coefficients sit in VGPRs ("v"): with "s" the 11 coefficient pairs push the kernel over the
// SGPR budget and come back as v_readlane + s_nop pairs (measured 0.411 ms vs 0.405 ms).
#ifndef PZ_COEF_CONSTRAINT
#define PZ_COEF_CONSTRAINT "v"
#endif
__device__ __forceinline__ double horner_step(double p, double r, double c)
{
    double out;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(out) : "v"(p), "v"(r), PZ_COEF_CONSTRAINT(c));
    return out;
}

__device__ __forceinline__ double fexp(double x)
{
#pragma clang fp contract(off)
    const double k = __builtin_rint(x * 1.4426950408889634);
    double r = fma(k, -0.6931471805599453, x);
    r = fma(k, -2.3190468138462996e-17, r);
    double p = horner_step(0x1.af395738f52d5p-26, r, 0x1.28923356dc7b1p-22);
    p = horner_step(p, r, 0x1.71de0d6feee62p-19);
    p = horner_step(p, r, 0x1.a019b87af85f3p-16);
    p = horner_step(p, r, 0x1.a01a01a7cf2d3p-13);
    p = horner_step(p, r, 0x1.6c16c178b1673p-10);
    p = horner_step(p, r, 0x1.11111111109b3p-7);
    p = horner_step(p, r, 0x1.5555555553d03p-5);
    p = horner_step(p, r, 0x1.5555555555556p-3);
    p = horner_step(p, r, 0x1.0000000000001p-1);
    p = fma(r * r, p, r);
    return ldexp(1.0 + p, (int)k);
}

// 2^t, <= 1 ulp, 15 VALU instructions: t = n + f with f = t - rint(t) exact, degree-11 polynomial
// 1 + f (c0 + c1 f + ... + c10 f^10) (Chebyshev-node interpolant of (2^f - 1)/f on |f| <= 1/2,
// max rel. error 1.14e-16 including evaluation rounding).  The kernels call it as
// fexp2(dtau * (-log2(e)/u)) with the per-angle constant formed on the host, which saves the
// argument multiply and the two-term ln2 reduction of fexp: exp(-dtau/u) then carries
// ~2 eps |dtau/u| of argument error instead of the reference's 1 eps |dtau/u| (np.exp(-dtau/u)).
// Saturating v_cvt_i32_f64 + v_ldexp_f64 give 0 / inf at the extremes; NaN propagates.
constexpr double NEG_LOG2E = -1.4426950408889634074;
__device__ __forceinline__ double fexp2(double t)
{
    const double n = __builtin_rint(t);
    const double f = t - n;
    double p = horner_step(0x1.e9d3fe3952179p-32, f, 0x1.e6063f7217bc6p-28);
    p = horner_step(p, f, 0x1.b524fae627834p-24);
    p = horner_step(p, f, 0x1.62bfd47773353p-20);
    p = horner_step(p, f, 0x1.ffcbfc670dcd4p-17);
    p = horner_step(p, f, 0x1.430913096fd9fp-13);
    p = horner_step(p, f, 0x1.5d87fe78a5276p-10);
    p = horner_step(p, f, 0x1.3b2ab6fba1ddap-7);
    p = horner_step(p, f, 0x1.c6b08d704a0c2p-5);
    p = horner_step(p, f, 0x1.ebfbdff82c598p-3);
    p = horner_step(p, f, 0x1.62e42fefa39efp-1);
    return ldexp(fma(f, p, 1.0), (int)n);
}

// The same polynomial with its coefficients held in VGPRs as ordinary (opaque) values: the compiler
// then selects the 3-address v_fma_f64 by itself (the coefficient is live across the loop, so the
// 2-address v_fmac_f64 is not an option) and, unlike around inline asm, it knows the instruction:
// no conservative s_nop after every Horner step (128 per layer in the 5-angle reflected kernel,
// ~3.7 cycles each when a wave runs alone on its SIMD; tools/ubench/f64_rates.hip).
struct Exp2Coef {
    double c[11];
    __device__ __forceinline__ void load()
    {
        const double k[11] = {0x1.62e42fefa39efp-1, 0x1.ebfbdff82c598p-3, 0x1.c6b08d704a0c2p-5,
                              0x1.3b2ab6fba1ddap-7, 0x1.5d87fe78a5276p-10, 0x1.430913096fd9fp-13,
                              0x1.ffcbfc670dcd4p-17, 0x1.62bfd47773353p-20, 0x1.b524fae627834p-24,
                              0x1.e6063f7217bc6p-28, 0x1.e9d3fe3952179p-32};
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            c[i] = k[i];
            asm volatile("" : "+v"(c[i]));      // opaque: not rematerialisable as a literal
        }
    }
};
__device__ __forceinline__ double fexp2(double t, const Exp2Coef &K)
{
#pragma clang fp contract(off)
    const double n = __builtin_rint(t);
    const double f = t - n;
    double p = fma(K.c[10], f, K.c[9]);
#pragma unroll
    for (int i = 8; i >= 0; --i) p = fma(p, f, K.c[i]);
    return ldexp(fma(f, p, 1.0), (int)n);
}

// fexp2 for the rarely taken side of a wave-uniform choice (direct exponential instead of a running
// product): the empty volatile asm keeps the compiler from if-converting the branch into a select
// that evaluates the polynomial on every layer.
__device__ __forceinline__ double fexp2_cold(double t, const Exp2Coef &K)
{
    asm volatile("" : "+v"(t));
    return fexp2(t, K);
}

// 1/b to ~1 ulp: v_rcp_f64 (2^-23 relative) + two Newton steps, 5 instructions
// (a correctly rounded a/b costs 11 with div_scale/div_fmas/div_fixup).
__device__ __forceinline__ double frcp(double b)
{
#pragma clang fp contract(off)
    double y = __builtin_amdgcn_rcp(b);
    double e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    return y;
}

// 1/b to 2^-46 relative (1.4e-14): v_rcp_f64 + ONE Newton step, 3 instructions.  Only where the reciprocal enters as
// a plain factor of a product (no difference is formed from the result): thermal emission's per-angle
// 1/((lam mu - 1)(lam mu + 1)), whose error multiplies the angle's source terms and nothing else; and, under PZ_REFL_DIET
// (round 5), the reflected solvers' per-angle 1/(den (lu - 1)(lu + 1)) (toon_reflected.hip, toon_reflected_coop.hip): a common
// factor of the layer's direct-beam terms and mode integrals, so the near-singular cancellation at lambda u0 -> 1 sees the
// same factor on both sides.  A deliberate 2^-46 against the reference's fp64 division there; -DPZ_REFL_DIET=0 restores the
// 1-ulp frcp, and tests/test_refl_coop_gpu.py / tools/headline_error_x87.py bound the difference.
__device__ __forceinline__ double frcp1(double b)
{
#pragma clang fp contract(off)
    const double y = __builtin_amdgcn_rcp(b);
    return fma(y, fma(-b, y, 1.0), y);
}

// sqrt(x), correctly rounded like the library call it replaces, for normal-range arguments: v_rsq_f64 seed,
// two coupled Goldschmidt steps on g ~ sqrt(x), h ~ 1/(2 sqrt(x)) and one residual correction
// g + (x - g g) h (exact residual in the fma); 12 instructions + the zero guard, against the compiler's
// expansion of sqrt() with its denormal scaling (two v_ldexp_f64, compare, selects) around the same core.
// Checked against the correctly rounded result on 1e5 random arguments (host emulation with a 2^-23 seed).
// sqrt(0) = 0 (conservative scattering: g1 = g2), negative arguments give NaN as sqrt() does.
__device__ __forceinline__ double fsqrt(double x)
{
#pragma clang fp contract(off)
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, x);
    g = fma(d, h, g);
    return (x == 0.0) ? x : g;
}

// a / b, correctly rounded like the division it replaces, for normal-range operands (no scaling, no fix-up of
// denormal or overflowing quotients: the callers' operands are physical constants over temperatures): v_rcp_f64
// seed, one Newton step, quotient, exact residual, correction -- 6 instructions against 11 for the
// v_div_scale / v_div_fmas / v_div_fixup sequence.  Checked like fsqrt.
__device__ __forceinline__ double fdiv(double a, double b)
{
#pragma clang fp contract(off)
    double y = __builtin_amdgcn_rcp(b);
    const double e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    const double q = a * y;
    const double r = fma(-b, q, a);
    return fma(r, y, q);
}

// Two-stream gamma coefficients, lambda and the direct-beam denominator-side quantities in the
// reference's exact (unfused) operation order (fluxes.py:1132-1141).  lambda^2 - 1/u0^2
// (fluxes.py:1155) is a true singularity of the particular solution: the reference's own result
// carries an error ~ eps / |lambda^2 - 1/u0^2| there, and any other rounding of lambda (for instance
// an FMA contraction of g1*g1 - g2*g2) shifts the answer by that much.  Forming these few values
// bit-for-bit like numpy keeps the GPU result on top of the reference even at near-singular
// (layer, wavelength, angle) points; everything downstream is insensitive to last-ulp changes.
__device__ __forceinline__ void toon_gammas(int toon_coefficients, double w0, double fcg,
                                            double &g1, double &g2, double &lam, double &lam2)
{
#pragma clang fp contract(off)
    if (toon_coefficients == 1) {                       // eddington (fluxes.py:1134-1135)
        g1 = (7.0 - w0 * (4.0 + 3.0 * fcg)) / 4.0;
        g2 = -(1.0 - w0 * (4.0 - 3.0 * fcg)) / 4.0;
    } else {                                            // quadrature (fluxes.py:1137-1138)
        g1 = (SQ3 * 0.5) * (2.0 - w0 * (1.0 + fcg));
        g2 = (SQ3 * w0 * 0.5) * (1.0 - fcg);
    }
    const double a = g1 * g1;
    const double b = g2 * g2;
    lam = fsqrt(a - b);                                 // fluxes.py:1140 (correctly rounded, as np.sqrt)
    lam2 = lam * lam;
}

// toon_gammas with ftau_cld*cosb = +0 folded by hand: 1 + 0 = 1, w0 * 1 = w0, 4 + 0 = 4 exactly, so
// the values are bit-identical to toon_gammas(.., fcg = 0, ..)
__device__ __forceinline__ void toon_gammas_nocld(int toon_coefficients, double w0, double &g1, double &g2,
                                                  double &lam, double &lam2)
{
#pragma clang fp contract(off)
    if (toon_coefficients == 1) {
        g1 = (7.0 - w0 * 4.0) / 4.0;
        g2 = -(1.0 - w0 * 4.0) / 4.0;
    } else {
        g1 = (SQ3 * 0.5) * (2.0 - w0);
        g2 = SQ3 * w0 * 0.5;
    }
    const double a = g1 * g1;
    const double b = g2 * g2;
    lam = fsqrt(a - b);
    lam2 = lam * lam;
}

__device__ __forceinline__ double sub_unfused(double a, double b)
{
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ double mul_unfused(double a, double b)
{
#pragma clang fp contract(off)
    return a * b;
}

// 1/sqrt(x) to ~1 ulp: v_rsq_f64 + two Newton steps (y <- y + y*(0.5 - 0.5 x y^2)), 9 instructions
// versus ~31 for sqrt() followed by a correctly rounded divide.
__device__ __forceinline__ double frsq(double x)
{
#pragma clang fp contract(off)
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    double e = fma(-(hx * y), y, 0.5);
    y = fma(y, e, y);
    e = fma(-(hx * y), y, 0.5);
    y = fma(y, e, y);
    return y;
}

// Henyey-Greenstein term in the frame of the downward beam:
// (1-g^2)/sqrt((1+g^2+2 g cos_theta)^3)   (reference picaso/fluxes.py:1308-1317)
__device__ __forceinline__ double hg_term(double g, double ct)
{
#pragma clang fp contract(off)
    const double b = fma(2.0 * g, ct, fma(g, g, 1.0));
    return fma(-g, g, 1.0) * frsq((b * b) * b);
}

// x**c with the common exponent 2 (config default TTHG fraction 1 - g_back^2) kept cheap.
__device__ __forceinline__ double pow_frac(double x, double c)
{
#pragma clang fp contract(off)
    return (c == 2.0) ? x * x : pow(x, c);
}

// Single-scattering phase function p_single for one layer.
//   1-D: reference picaso/fluxes.py:1303-1353 ; 3-D: reference picaso/fluxes.py:594-639
// (they differ only for single_phase == 0, 'cahoy').
template <bool IS3D>
__device__ __forceinline__ double p_single(int single_phase, double cosb_og, double gcos2,
                                           double ftau_cld, double ftau_ray, double ct,
                                           double frac_a, double frac_b, double frac_c,
                                           double constant_back, double constant_forward)
{
    // operations written out (explicit fma, no contraction): see reflected_layer
#pragma clang fp contract(off)
    if (single_phase == 1) return hg_term(cosb_og, ct);
    const double gf = constant_forward * cosb_og;
    const double gb = constant_back * cosb_og;
    const double f = fma(frac_b, pow_frac(gb, frac_c), frac_a);
    if (single_phase == 0) {
        if (!IS3D) return fma(1.0 - f, hg_term(gb, ct), fma(f, hg_term(gf, ct), gcos2));
        const double b1 = fma(2.0 * cosb_og, ct, fma(cosb_og, cosb_og, 1.0));
        const double hb = -cosb_og / 2.0;
        const double b2 = fma(2.0 * hb, ct, fma(hb, hb, 1.0));
        return fma((1.0 - f) * fma(-gb, gb, 1.0), frsq((b2 * b2) * b2),
                   fma(f * fma(-gf, gf, 1.0), frsq((b1 * b1) * b1), gcos2));
    }
    double tthg;
    if (!IS3D && ct == 1.0) {
        // zero phase angle (cos_theta = 1, the symmetric 1-D geometry): 1 + g^2 + 2g = (1+g)^2, so
        // (1-g^2)/((1+g)^2)^1.5 = (1-g)/(1+g)^2 -- one reciprocal for both terms instead of two rsqrt
        const double pf = 1.0 + gf, pb = 1.0 + gb;
        const double qf = pf * pf, qb = pb * pb;
        const double r = frcp(qf * qb);
        tthg = fma((1.0 - f) * (1.0 - gb), qf, (f * (1.0 - gf)) * qb) * r;
    } else {
        tthg = fma(1.0 - f, hg_term(gb, ct), f * hg_term(gf, ct));
    }
    if (single_phase == 2) return tthg;
    return fma(ftau_cld, tthg, ftau_ray * (0.75 * fma(ct, ct, 1.0)));
}

constexpr double LOG2E_D = 1.4426950408889634074;
// exp(x) through the base-2 polynomial with resident coefficients
__device__ __forceinline__ double fexpk(double x, const Exp2Coef &K) { return fexp2(x * LOG2E_D, K); }

// 1/(e - 1) of the Planck function.  Cold levels at short wavelengths overflow the exponential
// (hc wno/kT > 709.8: below ~68 K at 0.3 um): numpy forms 1/(inf - 1) = 0 there, while frcp(inf) is
// NaN (v_rcp gives 0 and the Newton step fma(-inf, 0, 1) is NaN), which would poison the disk sum,
// effective_temperature and fpfs_thermal of an otherwise finite spectrum.
__device__ __forceinline__ double planck_rcp(double e)
{
#pragma clang fp contract(off)
    const double y = frcp(e - 1.0);
    return (e == __builtin_inf()) ? 0.0 : y;
}

// Planck function per unit wavelength, cgs, at wavelength 1/wno (reference fluxes.py:1660-1680).
__device__ __forceinline__ double planck_lambda(double t, double wno)
{
#pragma clang fp contract(off)
    const double h = 6.62607004e-27, c = 2.99792458e+10, k = 1.38064852e-16;
    const double wcm = 1.0 / wno;
    const double w2 = wcm * wcm;
    return ((2.0 * h * (c * c)) / (w2 * w2 * wcm)) * planck_rcp(fexp(fdiv(h * c, t * (wcm * k))));
}
__device__ __forceinline__ double planck_lambda(double t, double wno, const Exp2Coef &K)
{
#pragma clang fp contract(off)
    const double h = 6.62607004e-27, c = 2.99792458e+10, k = 1.38064852e-16;
    const double wcm = 1.0 / wno;
    const double w2 = wcm * wcm;
    return ((2.0 * h * (c * c)) / (w2 * w2 * wcm)) * planck_rcp(fexpk(fdiv(h * c, t * (wcm * k)), K));
}

// 3-point bin mean of the wavenumber Planck function (reference fluxes.py:1608-1658, nbb = 1).
__device__ __forceinline__ double planck_integrated(double t, double wave, double dwave)
{
#pragma clang fp contract(off)
    const double h = 6.62607004e-27, c = 2.99792458e+10, k = 1.38064852e-16;
    const double c1 = 2 * h * (c * c), c2 = h * c / k;
    double s = 0.0;
#pragma unroll
    for (int kk = -1; kk <= 1; ++kk) {
        const double wn = wave + kk * dwave / 2.0;
        s += c1 * (wn * wn * wn) * planck_rcp(fexp(fdiv(c2 * wn, t)));
    }
    return s / 3.0;
}

}  // namespace pz
