// Level and mid-point fluxes (the climate caller's outputs) for the Toon89 two-stream solvers.
//
//   reflected: get_reflected_1d(..., get_lvl_flux=1)   reference picaso/fluxes.py:1219-1257
//   thermal  : get_thermal_1d, which always fills them  reference picaso/fluxes.py:1851-1907
//
// These need the solution (pos_i, neg_i) in every layer, so the single top-down sweep of the TOA
// kernels is followed by a bottom-up substitution: sweep 1 stores per layer the relation
// neg_i = delta_i - rho_i pos_i and the map pos_{i-1} = s_i pos_i + t_i (4 doubles per layer and
// column, in a context-owned scratch of 4 planes (nlayer, nwno)); the surface row fixes
// pos_{n-1}; sweep 2 walks up, recomputes the cheap per-layer coefficients from the input planes
// and writes the fluxes.  Same lane mapping as the TOA kernels (one lane per wavelength, every
// access a coalesced row segment); this path is bound by its own (numg,numt,nlevel,nwno) outputs.
#include "common.hpp"
#include "device_math.hpp"

namespace pz {

int lvl_scratch_reserve(picaso_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->lvl_scratch_bytes) return 0;
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->lvl_scratch) PZ_HIP(ctx, hipFree(ctx->lvl_scratch));
    ctx->lvl_scratch = nullptr;
    ctx->lvl_scratch_bytes = 0;
    PZ_HIP(ctx, hipMalloc((void **)&ctx->lvl_scratch, bytes));
    ctx->lvl_scratch_bytes = bytes;
    return 0;
}

int ck_scratch_reserve(picaso_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->ck_scratch_bytes) return 0;
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->ck_scratch) PZ_HIP(ctx, hipFree(ctx->ck_scratch));
    ctx->ck_scratch = nullptr;
    ctx->ck_scratch_bytes = 0;
    PZ_HIP(ctx, hipMalloc((void **)&ctx->ck_scratch, bytes));
    ctx->ck_scratch_bytes = bytes;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// reflected light, one angle
// ------------------------------------------------------------------------------------------------
struct ReflLayer {
    double g1, g2, lam, gam, E, EP, EM, am, ap;
};

__device__ __forceinline__ ReflLayer refl_layer_coeffs(const ReflectedArgs &a, long off, double F,
                                                       double u0, double iu0, double iu0sq)
{
    ReflLayer r;
    const double w0 = a.w0[off], fcg = a.ftau_cld[off] * a.cosb[off];
    double lam2;
    toon_gammas(a.toon_coefficients, w0, fcg, r.g1, r.g2, r.lam, lam2);
    r.gam = (r.g1 - r.lam) * frcp(r.g2);
    r.E = fmin(r.lam * a.dtau[off], 35.0);                  // fluxes.py:1172-1174
    r.EP = fexp(r.E);
    r.EM = frcp(r.EP);
    double g3;
    if (a.toon_coefficients == 1) g3 = (2.0 - 3.0 * fcg * u0) * 0.25;   // fluxes.py:1149
    else g3 = 0.5 * (1.0 - SQ3 * fcg * u0);                             // fluxes.py:1151
    const double g4 = 1.0 - g3;
    const double fw_den = F * w0 * frcp(sub_unfused(lam2, iu0sq));      // fluxes.py:1155-1159
    r.am = fw_den * (g4 * (r.g1 + iu0) + r.g2 * g3);
    r.ap = fw_den * (g3 * (r.g1 - iu0) + r.g2 * g4);
    return r;
}

__global__ __launch_bounds__(256) void k_reflected_lvl(const ReflectedLvlArgs A)
{
    const ReflectedArgs &a = A.base;
    const long w = blockIdx.x * (long)blockDim.x + threadIdx.x;     // column (wavelength x Gauss point)
    if (w >= a.ncol) return;
    const long wv = (a.ncolper > 1) ? w / a.ncolper : w;             // wavelength of this column
    const int n = a.nlayer;
    const long pitch = a.pitch, nw = a.ncol;
    const double u0 = a.ang[0].u0, iu0 = a.ang[0].iu0, iu0sq = a.ang[0].iu0sq;
    const double F = a.F0PI[wv], rs = a.surf_reflect[wv];
    double *s_rho = A.scratch + w, *s_del = s_rho + (long)n * nw, *s_s = s_del + (long)n * nw,
           *s_t = s_s + (long)n * nw;

    // ---- sweep 1: top-down elimination ----
    double rho = 0.0, delta = 0.0, pgam = 0.0, pEM = 0.0, pcpd = 0.0, pcmd = 0.0;
    double xu = fexp(-a.tau[w] * iu0);
    for (int i = 0; i < n; ++i) {
        const long off = (long)i * pitch + w;
        const ReflLayer r = refl_layer_coeffs(a, off, F, u0, iu0, iu0sq);
        const double xd = fexp(-a.tau[off + pitch] * iu0);
        const double cmu = r.am * xu, cpu = r.ap * xu, cmd = r.am * xd, cpd = r.ap * xd;
        xu = xd;
        double rho_n, delta_n, sfac = 0.0, t = 0.0;
        if (i == 0) {
            rho_n = r.gam;
            delta_n = a.b_top - cmu;                        // fluxes.py:155-158
        } else {
            const double em2 = pEM * pEM;
            const double a1 = 1.0 - pgam * em2 * rho, a2 = pgam - em2 * rho;
            const double inv = frcp(a1 - r.gam * a2);
            const double rP = (cpu - pcpd) - pgam * pEM * delta;
            const double rM = (cmu - pcmd) - pEM * delta;
            rho_n = (r.gam * a1 - a2) * inv;
            delta_n = (a2 * rP - a1 * rM) * inv;
            const double ia = pEM * frcp(a1);
            sfac = (1.0 - r.gam * rho_n) * ia;
            t = (r.gam * delta_n + rP) * ia;
        }
        s_rho[(long)i * nw] = rho_n;
        s_del[(long)i * nw] = delta_n;
        s_s[(long)i * nw] = sfac;
        s_t[(long)i * nw] = t;
        rho = rho_n; delta = delta_n; pgam = r.gam; pEM = r.EM; pcpd = cpd; pcmd = cmd;
    }
    // surface row (fluxes.py:178-183)
    const double tau_bot = a.tau[(long)n * pitch + w];
    const double xb = xu;                                   // exp(-tau[n]/u0)
    const double b_surface = 0.0 + rs * u0 * F * xb;
    const double em2b = pEM * pEM;
    double pos = (pEM * (b_surface - pcpd + rs * pcmd) - em2b * (pgam - rs) * delta) /
                 ((1.0 - rs * pgam) - em2b * (pgam - rs) * rho);
    (void)tau_bot;

    // ---- sweep 2: bottom-up substitution; pos_i replaces the sweep factor in the scratch ----
    for (int i = n - 1; i >= 0; --i) {
        const double pos_up = s_s[(long)i * nw] * pos + s_t[(long)i * nw];    // pos_{i-1}
        s_s[(long)i * nw] = pos;
        pos = pos_up;
    }
}

// Level and mid-layer fluxes from the two-stream solution (fluxes.py:1219-1257): one thread per
// (layer, column) -- the layers are independent once pos/neg are known, so this part of the work
// does not sit on the sequential sweep (it is most of the arithmetic: the coefficients and four
// exponentials per layer).
__global__ __launch_bounds__(256) void k_reflected_lvl_fluxes(const ReflectedLvlArgs A)
{
    const ReflectedArgs &a = A.base;
    const long w = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (w >= a.ncol) return;
    const long wv = (a.ncolper > 1) ? w / a.ncolper : w;
    const int n = a.nlayer, i = blockIdx.y;
    const long pitch = a.pitch, nw = a.ncol;
    const double u0 = a.ang[0].u0, iu0 = a.ang[0].iu0, iu0sq = a.ang[0].iu0sq;
    const double F = a.F0PI[wv];
    const double *s_rho = A.scratch + w, *s_del = s_rho + (long)n * nw, *s_pos = s_del + (long)n * nw;
    const double uF = u0 * F;
    const long off = (long)i * pitch + w;
    const ReflLayer r = refl_layer_coeffs(a, off, F, u0, iu0, iu0sq);
    const double pos = s_pos[(long)i * nw];
    const double neg = s_del[(long)i * nw] - s_rho[(long)i * nw] * pos;
    const double tau_i = a.tau[off], dt = a.dtau[off];
    const double xu_i = fexp(-tau_i * iu0);
    const double cmu = r.am * xu_i, cpu = r.ap * xu_i;
    const long o = (long)i * nw + w;
    A.fm[o] = (pos * r.gam + neg + cmu) + uF * xu_i;                     // :1227, :1236
    A.fp[o] = pos + r.gam * neg + cpu;                                    // :1228
    const double EPm = fexp(0.5 * r.E), EMm = frcp(EPm);                  // :1239-1240
    const double xm = fexp(-(tau_i + 0.5 * dt) * iu0);                    // :1243-1244
    A.fmm[o] = (r.gam * pos * EPm + neg * EMm + r.am * xm) + uF * xm;     // :1248, :1251
    A.fpm[o] = pos * EPm + r.gam * neg * EMm + r.ap * xm;                 // :1249
    if (i == n - 1) {                                                     // level n (:1230-1233)
        const double xd = fexp(-a.tau[off + pitch] * iu0);
        const long ob = (long)n * nw + w;
        A.fm[ob] = (r.gam * pos * r.EP + neg * r.EM + r.am * xd) + uF * xd;
        A.fp[ob] = pos * r.EP + r.gam * neg * r.EM + r.ap * xd;
        A.fmm[ob] = 0.0;
        A.fpm[ob] = 0.0;
    }
}

int launch_reflected_lvl(picaso_ctx *ctx, const ReflectedLvlArgs &a)
{
    const int block = a.base.ncol <= 64L * 256 ? 64 : 256;   // see launch_thermal_lvl
    const long grid = (a.base.ncol + block - 1) / block;
    hipLaunchKernelGGL(k_reflected_lvl, dim3((unsigned)grid), dim3(block), 0, ctx->stream, a);
    PZ_HIP(ctx, hipGetLastError());
    const long grid256 = (a.base.ncol + 255) / 256;
    hipLaunchKernelGGL(k_reflected_lvl_fluxes, dim3((unsigned)grid256, (unsigned)a.base.nlayer), dim3(256), 0,
                       ctx->stream, a);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// thermal emission, all angles (one solve per wavelength, fluxes.py:1812-1831)
// ------------------------------------------------------------------------------------------------
struct ThermLayer {
    double B0, b1, lam, gam, s, E, EP, EM, EPm, EMm, cmu, q;   // q = pi b1/(g1+g2): c+- = pi B +- q; EPm = exp(E/2)
};

__device__ __forceinline__ ThermLayer therm_layer_coeffs(const ThermalArgs &a, long off, double B0,
                                                         double Bn, const Exp2Coef &K)
{
    ThermLayer r;
    const double mu1 = 0.5;
    const double dt = a.dtau[off], w0 = a.w0[off], g = a.cosb[off];
    r.B0 = B0;
    r.b1 = (Bn - B0) * frcp(dt);                            // fluxes.py:1757
    const double g1 = 2.0 - w0 * (1 + g), g2 = w0 * (1 - g);
    r.lam = fsqrt(g1 * g1 - g2 * g2);
    r.gam = (g1 - r.lam) * frcp(g2);
    r.s = frcp(g1 + g2);
    // fluxes.py:1772-1779 with 2 pi mu1 = pi and B0 + b1 dtau = B_{i+1}:
    //   c+up = pi B_i + q, c-up = pi B_i - q, c+dn = pi B_{i+1} + q, c-dn = pi B_{i+1} - q
    r.q = PI * r.b1 * r.s;
    r.cmu = 2 * PI * mu1 * (B0 - r.b1 * r.s);
    r.E = fmin(r.lam * dt, 35.0);
    r.EPm = fexpk(0.5 * r.E, K);                            // exp(E/2) (fluxes.py:1856-1857) ...
    r.EMm = frcp(r.EPm);
    r.EP = r.EPm * r.EPm;                                   // ... and exp(E), exp(-E) from it
    r.EM = r.EMm * r.EMm;
    return r;
}

__global__ __launch_bounds__(256) void k_thermal_lvl_solve(const ThermalLvlArgs A)
{
    const ThermalArgs &a = A.base;
    const long w = blockIdx.x * (long)blockDim.x + threadIdx.x;     // column ([profile,] wavelength, Gauss point)
    if (w >= a.ncol) return;
    // several temperature profiles over one set of planes: wp = the column of the planes, item = the profile
    const long item = a.per_item ? w / a.per_item : 0, wp = a.per_item ? w - item * a.per_item : w;
    const long wv = (a.ncolper > 1) ? wp / a.ncolper : wp;
    const int n = a.nlayer;
    const long pitch = a.pitch, nw = a.ncol;
    const double mu1 = 0.5;
    const double wn = a.wno[wv], rs = a.surf_reflect[wv];
    const bool integrated = (a.calc_type == 1);
    const double dwn = integrated ? a.dwno[wv] : 0.0;
    Exp2Coef K;
    K.load();
    const double *tl = a.tlevel + item * (n + 1);
    auto planck = [&](int l) {
        const double t = tl[l];
        return integrated ? planck_integrated(t, wn, dwn) : planck_lambda(t, wn, K);
    };
    double *s_rho = A.scratch + w, *s_del = s_rho + (long)n * nw, *s_s = s_del + (long)n * nw,
           *s_t = s_s + (long)n * nw, *s_B = s_t + (long)n * nw;   // s_B: Planck function at the nlevel levels

    // ---- sweep 1 ----
    double rho = 0.0, delta = 0.0, pgam = 0.0, pEM = 0.0, pq = 0.0, b1_last = 0.0, s_last = 0.0;
    double Bn = planck(0);
    s_B[0] = Bn;
    const double B_top = Bn;
    const double tau_top = a.dtau[wp] * a.plevel[0] / (a.plevel[1] - a.plevel[0]);   // fluxes.py:1797
    for (int i = 0; i < n; ++i) {
        const double B0 = Bn;
        Bn = planck(i + 1);
        s_B[(long)(i + 1) * nw] = Bn;      // the per-angle passes below read it back (3 exp per level when integrated)
        const ThermLayer r = therm_layer_coeffs(a, (long)i * pitch + wp, B0, Bn, K);
        double rho_n, delta_n, sfac = 0.0, t = 0.0;
        if (i == 0) {
            rho_n = r.gam;
            delta_n = (1.0 - fexpk(-tau_top / mu1, K)) * B_top * PI - r.cmu;        // fluxes.py:1800
        } else {
            const double em2 = pEM * pEM;
            const double a1 = 1.0 - pgam * em2 * rho, a2 = pgam - em2 * rho;
            const double inv = frcp(a1 - r.gam * a2);
            // c+up_i - c+dn_{i-1} = q_i - q_{i-1} and c-up_i - c-dn_{i-1} = -(q_i - q_{i-1}) exactly:
            // the pi B terms cancel analytically instead of numerically
            const double dq = r.q - pq;
            const double rP = dq - pgam * pEM * delta;
            const double rM = -dq - pEM * delta;
            rho_n = (r.gam * a1 - a2) * inv;
            delta_n = (a2 * rP - a1 * rM) * inv;
            const double ia = pEM * frcp(a1);
            sfac = (1.0 - r.gam * rho_n) * ia;
            t = (r.gam * delta_n + rP) * ia;
        }
        s_rho[(long)i * nw] = rho_n;
        s_del[(long)i * nw] = delta_n;
        s_s[(long)i * nw] = sfac;
        s_t[(long)i * nw] = t;
        rho = rho_n; delta = delta_n; pgam = r.gam; pEM = r.EM; pq = r.q;
        b1_last = r.b1;
        s_last = r.s;
    }
    const double B_bot = Bn;
    // surface row: b_surface - c+dn + rs c-dn (fluxes.py:1802-1806, :181) with the pi B_n terms
    // cancelled analytically (the reference subtracts them numerically and loses up to 11 digits
    // in optically thick, weakly scattering bottom layers)
    const double bsum = a.hard_surface ? -PI * b1_last * s_last * (1.0 + rs)
                                       : PI * (b1_last * (mu1 - s_last) + rs * (B_bot - b1_last * s_last));
    const double em2b = pEM * pEM;
    double pos = (pEM * bsum - em2b * (pgam - rs) * delta) /
                 ((1.0 - rs * pgam) - em2b * (pgam - rs) * rho);
    // ---- sweep 2: pos/neg for every layer, kept in the first two scratch planes ----
    for (int i = n - 1; i >= 0; --i) {
        const double neg = s_del[(long)i * nw] - s_rho[(long)i * nw] * pos;
        const double pos_up = s_s[(long)i * nw] * pos + s_t[(long)i * nw];
        s_rho[(long)i * nw] = pos;
        s_del[(long)i * nw] = neg;
        pos = pos_up;
    }
}

// Per angle and direction (blockIdx.y = angle, blockIdx.z = 0 downward / 1 upward): Toon Table-3
// source-function sweeps (fluxes.py:1864-1910) on the two-stream solution k_thermal_lvl_solve left
// in the scratch planes (pos, neg per layer; Planck function per level).  Angles and directions are
// independent, so a correlated-k climate grid of a few thousand columns still fills the chip.
__global__ __launch_bounds__(256) void k_thermal_lvl_angle(const ThermalLvlArgs A)
{
    const ThermalArgs &a = A.base;
    const long w = blockIdx.x * (long)blockDim.x + threadIdx.x;     // column ([profile,] wavelength, Gauss point)
    if (w >= a.ncol) return;
    const long wp = a.per_item ? w % a.per_item : w;                // the column of the (shared) planes
    const long wv = (a.ncolper > 1) ? wp / a.ncolper : wp;
    const int n = a.nlayer, nlevel = n + 1;
    const long pitch = a.pitch, nw = a.ncol;
    const double mu1 = 0.5;
    const double rs = a.surf_reflect[wv];
    Exp2Coef K;
    K.load();
    const double *s_rho = A.scratch + w, *s_del = s_rho + (long)n * nw, *s_B = s_del + 3 * (long)n * nw;
    const int k = blockIdx.y;
    const bool upward = blockIdx.z == 1;
    const double mu = A.u1_dev[k], imu = 1.0 / mu, nlh = 0.5 * NEG_LOG2E * imu;   // exp(-x/(2 mu)) = 2^(x nlh)
    double *fm = A.fm + ((long)k * nlevel) * nw + w, *fp = A.fp + ((long)k * nlevel) * nw + w,
           *fmm = A.fmm + ((long)k * nlevel) * nw + w, *fpm = A.fpm + ((long)k * nlevel) * nw + w;
    if (!upward) {
        const double B_top = s_B[0];
        const double tau_top = a.dtau[wp] * a.plevel[0] / (a.plevel[1] - a.plevel[0]);   // fluxes.py:1797
        double Bcur = B_top;
        double Fm = (1 - fexpk(-tau_top * imu, K)) * B_top * 2 * PI;                 // :1875
        fm[0] = Fm;
        for (int i = 0; i < n; ++i) {
            const double B0 = Bcur;
            Bcur = s_B[(long)(i + 1) * nw];
            const long off = (long)i * pitch + wp;
            const ThermLayer r = therm_layer_coeffs(a, off, B0, Bcur, K);
            const double dt = a.dtau[off];
            const double P = s_rho[(long)i * nw], N = s_del[(long)i * nw];
            const double J = r.gam * (r.lam + 1.0 / mu1) * P, Kc = (1.0 / mu1 - r.lam) * N;  // :1844-1845
            const double si1 = 2 * PI * (r.B0 - r.b1 * (r.s - mu1)), si2 = 2 * PI * r.b1;     // :1848-1849
            const double eam = fexp2(dt * nlh, K), ea = eam * eam;        // exp(-dtau/(2 mu)), exp(-dtau/mu)
            const double EPm = r.EPm, EMm = r.EMm;
            const double lp1 = r.lam * mu + 1.0, lm1 = r.lam * mu - 1.0, r2 = frcp(lp1 * lm1);
            const double lup = r2 * lm1, lum = r2 * lp1;                  // 1/(lam mu + 1), 1/(lam mu - 1)
            fmm[(long)i * nw] = (Fm * eam + (J * lup) * (EPm - eam) - (Kc * lum) * (EMm - eam) +
                                 si1 * (1. - eam) + si2 * (mu * eam + 0.5 * dt - mu));         // :1889-1893
            Fm = (Fm * ea + (J * lup) * (r.EP - ea) + (Kc * lum) * (ea - r.EM) + si1 * (1. - ea) +
                  si2 * (mu * ea + dt - mu));                                                  // :1883-1887
            fm[(long)(i + 1) * nw] = Fm;
        }
        fmm[(long)n * nw] = 0.0;
        return;
    }
    const double Bb = s_B[(long)n * nw];
    // b1 of the bottom layer, formed as therm_layer_coeffs forms it
    const double b1_last = (Bb - s_B[(long)(n - 1) * nw]) * frcp(a.dtau[(long)(n - 1) * pitch + wp]);
    double Fp = a.hard_surface ? (1.0 - rs) * Bb * 2 * PI : (Bb + b1_last * mu) * 2 * PI;  // :1871-1873
    fp[(long)n * nw] = Fp;
    fpm[(long)n * nw] = 0.0;
    double Bnext = Bb;
    for (int i = n - 1; i >= 0; --i) {
        const double B0 = s_B[(long)i * nw];
        const long off = (long)i * pitch + wp;
        const ThermLayer r = therm_layer_coeffs(a, off, B0, Bnext, K);
        Bnext = B0;
        const double dt = a.dtau[off];
        const double P = s_rho[(long)i * nw], N = s_del[(long)i * nw];
        const double G = (1.0 / mu1 - r.lam) * P, H = r.gam * (r.lam + 1.0 / mu1) * N;    // :1842-1843
        const double al1 = 2 * PI * (r.B0 + r.b1 * (r.s - mu1)), al2 = 2 * PI * r.b1;     // :1846-1847
        const double eam = fexp2(dt * nlh, K), ea = eam * eam;
        const double EPm = r.EPm, EMm = r.EMm;
        const double lp1 = r.lam * mu + 1.0, lm1 = r.lam * mu - 1.0, r2 = frcp(lp1 * lm1);
        const double lup = r2 * lm1, lum = r2 * lp1;
        fpm[(long)i * nw] = (Fp * eam + (G * lum) * (r.EP * eam - EPm) - (H * lup) * (r.EM * eam - EMm) +
                             al1 * (1. - eam) + al2 * (mu + 0.5 * dt - (dt + mu) * eam)); // :1903-1907
        Fp = (Fp * ea + (G * lum) * (r.EP * ea - 1.0) + (H * lup) * (1.0 - r.EM * ea) + al1 * (1. - ea) +
              al2 * (mu - (dt + mu) * ea));                                                // :1897-1901
        fp[(long)i * nw] = Fp;
    }
    A.flux[(long)k * nw + w] = fpm[0];                                                     // :1910
}

int launch_thermal_lvl(picaso_ctx *ctx, const ThermalLvlArgs &a)
{
    // one wave per block while the column count is small: the sweeps are latency bound and a
    // correlated-k climate grid (a few thousand columns) then spreads over as many CUs as it has waves
    const int block = a.base.ncol <= 64L * 256 ? 64 : 256;
    const long grid = (a.base.ncol + block - 1) / block;
    hipLaunchKernelGGL(k_thermal_lvl_solve, dim3((unsigned)grid), dim3(block), 0, ctx->stream, a);
    PZ_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_thermal_lvl_angle, dim3((unsigned)grid, (unsigned)a.nang, 2u), dim3(block), 0, ctx->stream, a);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace pz
