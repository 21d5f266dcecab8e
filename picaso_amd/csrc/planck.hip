// The reference's two public Planck-function tables -- gfx950.
//
// fluxes.blackbody(t, w) (reference picaso/fluxes.py:1660-1680: per unit wavelength, cgs, w in cm, result
// (ntemp, nwave)) and fluxes.blackbody_integrated(T, wave, dwave) (:1609-1658: the 3-point bin mean of the
// wavenumber Planck function the climate solver's thermal calls use, result (ntemp, nwave)).  The solvers never
// materialise these tables (toon_thermal.hip evaluates the same device functions level by level inside the sweep);
// the entry points exist because callers of the reference use them on their own (justplotit.py:976, 1610; brightness
// temperatures; fluxes.py:1752-1754 is `all_b = blackbody(tlevel, 1/wno)`).  Element-wise, one lane per (T, wave)
// pair with the wavelength index fastest: the write is the only HBM traffic that matters (8 B per element).
#include "common.hpp"
#include "device_math.hpp"

namespace pz {

// planck_lambda of device_math.hpp takes a wavenumber and forms wcm = 1/wno; the public function is handed the
// wavelength itself (the reference's thermal call passes 1/wno, so both see the same wcm bits)
__device__ __forceinline__ double planck_lambda_cm(double t, double wcm)
{
#pragma clang fp contract(off)
    const double h = 6.62607004e-27, c = 2.99792458e+10, k = 1.38064852e-16;
    const double w2 = wcm * wcm;
    return ((2.0 * h * (c * c)) / (w2 * w2 * wcm)) * planck_rcp(fexp(fdiv(h * c, t * (wcm * k))));
}

template <bool INTEGRATED>
__global__ __launch_bounds__(256) void k_blackbody(int ntemp, long nwave, const double *__restrict__ t,
                                                   const double *__restrict__ w, const double *__restrict__ dw,
                                                   double *__restrict__ out)
{
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= nwave) return;
    const double wi = w[i], dwi = INTEGRATED ? dw[i] : 0.0;
    for (int j = blockIdx.y; j < ntemp; j += gridDim.y) {
        const double tj = t[j];
        out[(size_t)j * nwave + i] = INTEGRATED ? planck_integrated(tj, wi, dwi) : planck_lambda_cm(tj, wi);
    }
}

static int launch_blackbody(picaso_ctx *ctx, bool integrated, int ntemp, long nwave, const double *t, const double *w,
                            const double *dw, double *out)
{
    if (ntemp < 1 || nwave < 1) return fail(ctx, "blackbody: ntemp and nwave must be positive");
    if (!t || !w || !out || (integrated && !dw)) return fail(ctx, "blackbody: null argument");
    const dim3 grid((unsigned)((nwave + 255) / 256), (unsigned)(ntemp < 1024 ? ntemp : 1024));
    if (integrated) hipLaunchKernelGGL(k_blackbody<true>, grid, dim3(256), 0, ctx->stream, ntemp, nwave, t, w, dw, out);
    else hipLaunchKernelGGL(k_blackbody<false>, grid, dim3(256), 0, ctx->stream, ntemp, nwave, t, w, dw, out);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

static int blackbody_host(picaso_ctx *ctx, bool integrated, int ntemp, const double *t, long nwave, const double *w,
                          const double *dw, double *out)
{
    if (!ctx) return fail(nullptr, "null context");
    if (ntemp < 1 || nwave < 1) return fail(ctx, "blackbody: ntemp and nwave must be positive");
    if (!t || !w || !out || (integrated && !dw)) return fail(ctx, "blackbody: null argument");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nout = (size_t)ntemp * (size_t)nwave;
    PZ_TRY(arena_reset(ctx, sizeof(double) * (nout + (size_t)ntemp + 2 * (size_t)nwave) + 16 * 256));
    const double *d_t, *d_w, *d_dw = nullptr;
    PZ_TRY(arena_upload(ctx, t, (size_t)ntemp, &d_t));
    PZ_TRY(arena_upload(ctx, w, (size_t)nwave, &d_w));
    if (integrated) PZ_TRY(arena_upload(ctx, dw, (size_t)nwave, &d_dw));
    double *d_o = (double *)arena_take(ctx, sizeof(double) * nout);
    if (!d_o) return fail(ctx, "arena exhausted");
    PZ_TRY(launch_blackbody(ctx, integrated, ntemp, nwave, d_t, d_w, d_dw, d_o));
    PZ_HIP(ctx, hipMemcpyAsync(out, d_o, sizeof(double) * nout, hipMemcpyDeviceToHost, ctx->stream));
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // namespace pz

using namespace pz;

extern "C" {

int picaso_blackbody(picaso_ctx *ctx, int ntemp, const double *t, long nwave, const double *w_cm, double *out)
{
    return blackbody_host(ctx, false, ntemp, t, nwave, w_cm, nullptr, out);
}

int picaso_blackbody_dev(picaso_ctx *ctx, int ntemp, const double *t, long nwave, const double *w_cm, double *out)
{
    if (!ctx) return fail(nullptr, "null context");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    return launch_blackbody(ctx, false, ntemp, nwave, t, w_cm, nullptr, out);
}

int picaso_blackbody_integrated(picaso_ctx *ctx, int ntemp, const double *T, long nwave, const double *wave,
                                const double *dwave, double *out)
{
    return blackbody_host(ctx, true, ntemp, T, nwave, wave, dwave, out);
}

int picaso_blackbody_integrated_dev(picaso_ctx *ctx, int ntemp, const double *T, long nwave, const double *wave,
                                    const double *dwave, double *out)
{
    if (!ctx) return fail(nullptr, "null context");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    return launch_blackbody(ctx, true, ntemp, nwave, T, wave, dwave, out);
}

}  // extern "C"
