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
#include "common.hpp"
#include "device_math.hpp"

namespace pz {

template <int NA, bool IS3D>
__global__ __launch_bounds__(256) void k_thermal_toa(const ThermalArgs a)
{
    // Operations written out (no contraction, explicit fma), as in toon_reflected.hip: the one-angle
    // launch of a small wavelength shard and the five-angle launch of the whole grid round identically,
    // so a column's flux does not depend on how the grid is cut.
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
        u1[k] = IS3D ? a.u1_tab[fac] : a.u1[blockIdx.y * NA + k];   // 3-D thermal takes ubar1 as is
        nl1[k] = NEG_LOG2E / u1[k];
    }
    const double *p_dtau = a.dtau + col, *p_w0 = a.w0 + col, *p_cosb = a.cosb + col;

    auto planck = [&](int l) {
        const double t = tl[(long)l * lstride];
        return integrated ? planck_integrated(t, wn, dwn) : planck_lambda(t, wn, K);
    };

    double W[NA], kappa[NA], zeta[NA];
    double rho = 0.0, delta = 0.0, pgam = 0.0, pEM = 0.0, pq = 0.0, b1_last = 0.0, s_last = 0.0;
    double Bn = planck(0);
    const double B_top = Bn;
    double tau_top = 0.0;

    double n_dt = p_dtau[0], n_w0 = p_w0[0], n_cb = p_cosb[0];
    for (int i = 0; i < n; ++i) {
        const double dt = n_dt, w0 = n_w0, g = n_cb;
        if (i + 1 < n) {
            const long o = (long)(i + 1) * pitch;
            n_dt = p_dtau[o];
            n_w0 = p_w0[o];
            n_cb = p_cosb[o];
        }
        const double B0 = Bn;
        Bn = planck(i + 1);
        const double b1 = (Bn - B0) * frcp(dt);                // fluxes.py:1757
        const double g1 = 2.0 - w0 * (1 + g), g2 = w0 * (1 - g);   // fluxes.py:1760
        const double lam = sqrt(g1 * g1 - g2 * g2);                // unfused, as numpy
        const double gam = (g1 - lam) * frcp(g2);
        const double s = frcp(g1 + g2);                        // fluxes.py:1766
        // fluxes.py:1772-1779 with 2 pi mu1 = pi and B0 + b1 dtau = B_{i+1}:
        //   c+up = pi B_i + q, c-up = pi B_i - q, c+dn = pi B_{i+1} + q, c-dn = pi B_{i+1} - q,
        // so the interface right-hand sides are +-(q_i - q_{i-1}) with the pi B terms cancelled
        // analytically (the reference cancels them numerically, at a cost of up to 11 digits in
        // optically thick, weakly scattering layers).
        const double q = (PI * b1) * s;
        const double cmu = (2 * PI * mu1) * fma(-b1, s, B0);
        const double E = fmin(lam * dt, 35.0);                 // fluxes.py:1784-1786
        const double EP = fexpk(E, K), EM = frcp(EP);
        const double al1 = (2 * PI) * fma(b1, s - mu1, B0);    // fluxes.py:1846-1847
        const double al2 = 2 * PI * b1;
        const double gcoef = (1.0 / mu1 - lam);                // G = gcoef*pos   fluxes.py:1842
        const double hcoef = gam * (lam + 1.0 / mu1);          // H = hcoef*neg   fluxes.py:1843

        double rho_n = gam, delta_n = 0.0, sfac = 0.0, t = 0.0;
        if (i == 0) {
            tau_top = dt * pl[0] / (pl[lstride] - pl[0]);      // fluxes.py:1797
            const double b_top = IS3D ? PI * (1.0 - fexpk(-tau_top / mu1, K)) * B_top   // :2253
                                      : (1.0 - fexpk(-tau_top / mu1, K)) * B_top * PI;  // :1800
            delta_n = b_top - cmu;
        } else {
            const double em2 = pEM * pEM;
            const double a1 = fma(-(pgam * em2), rho, 1.0);
            const double a2 = fma(-em2, rho, pgam);
            const double d1 = fma(-gam, a2, a1);
            const double r12 = frcp(d1 * a1);                  // one reciprocal for 1/d1 and 1/a1
            const double inv = r12 * a1;
            const double dq = q - pq;
            const double rP = fma(-(pgam * pEM), delta, dq);
            const double rM = fma(-pEM, delta, -dq);
            rho_n = fma(gam, a1, -a2) * inv;
            delta_n = fma(a2, rP, -(a1 * rM)) * inv;
            const double ia = pEM * (r12 * d1);
            sfac = fma(-gam, rho_n, 1.0) * ia;
            t = fma(gam, delta_n, rP) * ia;
        }
        const bool last = (i == n - 1);
        double EPm = 0.0, EMm = 0.0;
        if (i == 0) {
            EPm = fexpk(0.5 * E, K);                           // fluxes.py:1856-1857
            EMm = frcp(EPm);
        }
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const double mu = u1[k];
            // 1/(lam mu - 1) and 1/(lam mu + 1) from one reciprocal of the product of the two
            // 1-ulp factors (see toon_reflected.hip: no cancellation, lm1 is exact near lam mu = 1)
            const double lmu = lam * mu;
            const double lm1 = lmu - 1.0, lp1 = lmu + 1.0;
            const double r2 = frcp(lm1 * lp1);
            const double lp = gcoef * (r2 * lp1), lm = hcoef * (r2 * lm1);
            if (i == 0) {
                const double em = fexp2((0.5 * dt) * nl1[k], K); // fluxes.py:1878
                const double vp = lp * fma(EP, em, -EPm);      // fluxes.py:1903-1907
                const double vn = -lm * fma(EM, em, -EMm);
                const double c0 = fma(al2, fma(-(dt + mu), em, mu + 0.5 * dt), al1 * (1. - em));
                kappa[k] = fma(vn, delta_n, c0);
                zeta[k] = fma(-vn, rho_n, vp);
                W[k] = em;
            } else {
                const double e = fexp2(dt * nl1[k], K);        // fluxes.py:1877
                const double vp = (W[k] * lp) * fma(EP, e, -1.0);  // fluxes.py:1897-1901
                const double vn = (W[k] * lm) * fma(-EM, e, 1.0);
                const double c0 = W[k] * fma(al2, fma(-(dt + mu), e, mu), al1 * (1. - e));
                kappa[k] = fma(vn, delta_n, fma(zeta[k], t, kappa[k] + c0));
                zeta[k] = fma(-vn, rho_n, fma(zeta[k], sfac, vp));
                W[k] = W[k] * e;
            }
            if (last) {                                        // F+[n] boundary intensity
                double fb;
                if (!IS3D) {
                    if (a.hard_surface) fb = ((1.0 - rs) * Bn) * (2 * PI);   // fluxes.py:1871
                    else fb = fma(b1, mu, Bn) * (2 * PI);                    // fluxes.py:1873
                } else {
                    if (a.hard_surface) fb = PI * (PI * Bn);                 // fluxes.py:2256,2310
                    else fb = PI * fma(b1, mu, Bn);                          // fluxes.py:2312
                }
                // for a single layer the boundary feeds the mid-point of layer 0 directly
                kappa[k] = fma(W[k], fb, kappa[k]);
            }
        }
        rho = rho_n;
        delta = delta_n;
        pgam = gam;
        pEM = EM;
        pq = q;
        b1_last = b1;
        s_last = s;
    }
    // surface row b_surface - c+dn + rs c-dn with the pi B_n terms cancelled analytically:
    //   1-D soft (fluxes.py:1806) : pi [b1 (mu1 - s) + rs (B_n - b1 s)]
    //   1-D hard (fluxes.py:1803) : -pi b1 s (1 + rs)                (b_surface = (1-rs) pi B_n)
    //   3-D soft (fluxes.py:2258) : same as 1-D soft
    //   3-D hard (fluxes.py:2256) : pi [rs B_n - b1 s (1 + rs)]     (b_surface = pi B_n, no emissivity)
    double bsum;
    const double bs_l = b1_last * s_last;
    if (!a.hard_surface) bsum = PI * fma(rs, Bn - bs_l, b1_last * (mu1 - s_last));
    else if (!IS3D) bsum = -PI * (bs_l * (1.0 + rs));
    else bsum = PI * fma(rs, Bn, -(bs_l * (1.0 + rs)));
    const double em2 = pEM * pEM;
    const double egr = em2 * (pgam - rs);
    const double pos = fma(pEM, bsum, -(egr * delta)) / fma(-egr, rho, fma(-rs, pgam, 1.0));
    double disk = 0.0;
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const double x = fma(zeta[k], pos, kappa[k]);
        if (IS3D) a.flux[(long)fac * a.nwno + w] = x;
        else a.flux[(long)(blockIdx.y * NA + k) * a.ncol + col] = x;
        {   // flux + x*gweight*tweight in the reference's order, unfused (disco.py:174-176), as k_compress
#pragma clang fp contract(off)
            const int ia = IS3D ? 0 : blockIdx.y * NA + k;
            disk = disk + x * a.wgt[ia] * a.wgt2[ia];
        }
    }
    if (!IS3D && a.disk) {                                     // fused disco.compress_thermal
        double acc = a.disk_first ? disk : a.disk[w] + disk;
        if (a.disk_last) acc = acc * a.disk_scale;
        a.disk[w] = acc;
    }
}

template <int NA>
static int launch1d(picaso_ctx *ctx, const ThermalArgs &a)
{
    const int block = 256;
    const dim3 grid((unsigned)((a.ncol + block - 1) / block), (unsigned)(a.ny > 1 ? a.ny : 1));
    hipLaunchKernelGGL((k_thermal_toa<NA, false>), grid, dim3(block), 0, ctx->stream, a);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

int launch_thermal_toa(picaso_ctx *ctx, const ThermalArgs &a, bool is3d)
{
    if (a.ncol <= 0 || a.nlayer < 1) return fail(ctx, "thermal: empty problem");
    if (is3d) {
        const int block = 256;
        const long grid = (a.ncol + block - 1) / block;
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
