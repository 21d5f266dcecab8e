// Device-side helpers shared by the Toon kernels (gfx950, fp64).
#pragma once
#include <hip/hip_runtime.h>

namespace pz {

constexpr double PI = 3.14159265358979323846;
constexpr double SQ3 = 1.7320508075688772935;

// Henyey-Greenstein term in the frame of the downward beam:
// (1-g^2)/sqrt((1+g^2+2 g cos_theta)^3)   (reference picaso/fluxes.py:1308-1317)
__device__ __forceinline__ double hg_term(double g, double ct)
{
    const double b = 1.0 + g * g + 2.0 * g * ct;
    return (1.0 - g * g) / sqrt(b * b * b);
}

// x**c with the common exponent 2 (config default TTHG fraction 1 - g_back^2) kept cheap.
__device__ __forceinline__ double pow_frac(double x, double c)
{
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
    if (single_phase == 1) return hg_term(cosb_og, ct);
    const double gf = constant_forward * cosb_og;
    const double gb = constant_back * cosb_og;
    const double f = frac_a + frac_b * pow_frac(gb, frac_c);
    if (single_phase == 0) {
        if (!IS3D) return f * hg_term(gf, ct) + (1.0 - f) * hg_term(gb, ct) + gcos2;
        const double b1 = 1.0 + cosb_og * cosb_og + 2.0 * cosb_og * ct;
        const double hb = -cosb_og / 2.0;
        const double b2 = 1.0 + hb * hb + 2.0 * hb * ct;
        return f * (1.0 - gf * gf) / sqrt(b1 * b1 * b1) +
               (1.0 - f) * (1.0 - gb * gb) / sqrt(b2 * b2 * b2) + gcos2;
    }
    const double tthg = f * hg_term(gf, ct) + (1.0 - f) * hg_term(gb, ct);
    if (single_phase == 2) return tthg;
    return ftau_cld * tthg + ftau_ray * (0.75 * (1.0 + ct * ct));
}

// Planck function per unit wavelength, cgs, at wavelength 1/wno (reference fluxes.py:1660-1680).
__device__ __forceinline__ double planck_lambda(double t, double wno)
{
    const double h = 6.62607004e-27, c = 2.99792458e+10, k = 1.38064852e-16;
    const double wcm = 1.0 / wno;
    const double w2 = wcm * wcm;
    return ((2.0 * h * (c * c)) / (w2 * w2 * wcm)) * (1.0 / (exp((h * c) / (t * (wcm * k))) - 1.0));
}

// 3-point bin mean of the wavenumber Planck function (reference fluxes.py:1608-1658, nbb = 1).
__device__ __forceinline__ double planck_integrated(double t, double wave, double dwave)
{
    const double h = 6.62607004e-27, c = 2.99792458e+10, k = 1.38064852e-16;
    const double c1 = 2 * h * (c * c), c2 = h * c / k;
    double s = 0.0;
#pragma unroll
    for (int kk = -1; kk <= 1; ++kk) {
        const double wn = wave + kk * dwave / 2.0;
        s += c1 * (wn * wn * wn) / (exp(c2 * wn / t) - 1.0);
    }
    return s / 3.0;
}

}  // namespace pz
