// Gauss-Chebyshev disk quadrature -- gfx950.
// Replaces disco.compress_disco / compress_thermal (reference picaso/disco.py:117-181).
// One lane per output element; the (g,t) sum runs in the reference's loop order so the result
// is deterministic and independent of how the wavelength grid is sharded.
#include "common.hpp"

namespace pz {


// The two disk integrals differ only in how the (g,t) sum is finished:
//   COMPRESS_DISCO   : sym_fac*0.5*albedo/F0PI*(cos_theta+1)   (disco.py:148)   c1 = sym_fac*0.5, c2 = cos_theta+1
//   COMPRESS_THERMAL : flux*sym_fac                           (disco.py:181)   c1 = sym_fac
__device__ __forceinline__ double compress_finish(int mode, double acc, double c1, double c2, const double *F0PI,
                                                  size_t w)
{
#pragma clang fp contract(off)
    return (mode == COMPRESS_DISCO) ? c1 * acc / (F0PI ? F0PI[w] : 1.0) * c2 : acc * c1;   // F0PI == NULL: F0PI = 1
}

__global__ __launch_bounds__(256) void k_compress(size_t ninner, const double *__restrict__ x,
                                                  const double *__restrict__ wts, int nang,
                                                  const double *__restrict__ F0PI, int mode, double c1,
                                                  double c2, double *__restrict__ out)
{
    const size_t w = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (w >= ninner) return;
    double acc = 0.0;
    {
#pragma clang fp contract(off)
        for (int k = 0; k < nang; ++k) acc = acc + x[(size_t)k * ninner + w] * wts[2 * k] * wts[2 * k + 1];
    }
    out[w] = compress_finish(mode, acc, c1, c2, F0PI, w);
}

// Up to COMPRESS_ARG_ANGLES (g,t) weight pairs travel as kernel arguments: no table upload (a host-to-
// device copy on the stream ahead of the launch) for a kernel that lasts a few microseconds.
constexpr int COMPRESS_ARG_ANGLES = 128;
struct CompressArgs {
    size_t ninner;
    const double *x, *F0PI;
    double *out;
    int nang, mode;
    double c1, c2;
    double wts[2 * COMPRESS_ARG_ANGLES];
};
__global__ __launch_bounds__(256) void k_compress_args(const CompressArgs a)
{
    const size_t w = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (w >= a.ninner) return;
    double acc = 0.0;
    {
#pragma clang fp contract(off)
        for (int k = 0; k < a.nang; ++k) acc = acc + a.x[(size_t)k * a.ninner + w] * a.wts[2 * k] * a.wts[2 * k + 1];
    }
    a.out[w] = compress_finish(a.mode, acc, a.c1, a.c2, a.F0PI, w);
}

// host weight pairs (nang <= COMPRESS_ARG_ANGLES)
int launch_compress_hostw(picaso_ctx *ctx, size_t ninner, const double *x, const double *wts_host, int nang,
                          const double *F0PI, int mode, double c1, double c2, double *out)
{
    if (ninner == 0) return 0;
    if (nang < 1 || nang > COMPRESS_ARG_ANGLES)
        return fail(ctx, "disk integration: %d (g,t) weight pairs do not fit the %d carried as kernel arguments",
                    nang, COMPRESS_ARG_ANGLES);
    CompressArgs a{};
    a.ninner = ninner; a.x = x; a.F0PI = F0PI; a.out = out; a.nang = nang; a.mode = mode; a.c1 = c1; a.c2 = c2;
    for (int k = 0; k < 2 * nang; ++k) a.wts[k] = wts_host[k];
    const int block = 256;
    hipLaunchKernelGGL(k_compress_args, dim3((unsigned)((ninner + block - 1) / block)), dim3(block), 0, ctx->stream, a);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

// wts_dev: device table of nang (gweight[g], tweight[t]) pairs in (g,t) loop order
int launch_compress_dev(picaso_ctx *ctx, size_t ninner, const double *x, const double *wts_dev,
                        int nang, const double *F0PI, int mode, double c1, double c2, double *out)
{
    if (ninner == 0) return 0;
    const int block = 256;
    const size_t grid = (ninner + block - 1) / block;
    hipLaunchKernelGGL(k_compress, dim3((unsigned)grid), dim3(block), 0, ctx->stream, ninner, x,
                       wts_dev, nang, F0PI, mode, c1, c2, out);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}


// Correlated-k Gauss-point sum (reference justdoit.py:307, 380: `xint_at_top += xint*gauss_wts[ig]`,
// in ig order): in is (nrows, nwno, n) with the Gauss index fastest, out (nrows, nwno).
struct ColsumArgs {
    long nwno, nrows;
    int n;
    double wts[MAX_CK_GAUSS];
    const double *in;
    double *out;
};

__global__ __launch_bounds__(256) void k_weighted_colsum(const ColsumArgs a)
{
#pragma clang fp contract(off)      // `xint_at_top += xint*gauss_wts[ig]`: a product and a sum, as numpy
    const long w = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (w >= a.nwno) return;
    // rows (angle x level of a level-flux batch) can exceed the 65 535 limit of grid.y: stride over them
    for (long row = blockIdx.y; row < a.nrows; row += gridDim.y) {
        const double *src = a.in + (row * a.nwno + w) * a.n;
        double acc = 0.0;
        for (int j = 0; j < a.n; ++j) acc = acc + src[j] * a.wts[j];
        a.out[row * a.nwno + w] = acc;
    }
}

int launch_weighted_colsum(picaso_ctx *ctx, int nrows, long nwno, int n, const double *wts_host,
                           const double *in, double *out)
{
    if (n < 1 || n > MAX_CK_GAUSS) return fail(ctx, "gauss sum: ngauss must be 1..%d, got %d", MAX_CK_GAUSS, n);
    if (nrows < 1 || nwno < 1) return 0;
    ColsumArgs a{};
    a.nwno = nwno; a.n = n; a.in = in; a.out = out;
    for (int j = 0; j < n; ++j) a.wts[j] = wts_host[j];
    const int block = 256;
    a.nrows = nrows;
    hipLaunchKernelGGL(k_weighted_colsum, dim3((unsigned)((nwno + block - 1) / block), (unsigned)(nrows < 65535 ? nrows : 65535)),
                       dim3(block), 0, ctx->stream, a);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

// out may alias x or y (the patchy-cloud blends are in place): no __restrict__ here
__global__ __launch_bounds__(256) void k_axpby(size_t n, double a, const double *x, double b, const double *y,
                                               double *out)
{
#pragma clang fp contract(off)      // two products and a sum, as numpy evaluates the blend
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = a * x[i] + b * y[i];
}

int launch_axpby(picaso_ctx *ctx, size_t n, double a, const double *x, double b, const double *y, double *out)
{
    if (n == 0) return 0;
    const int block = 256;
    hipLaunchKernelGGL(k_axpby, dim3((unsigned)((n + block - 1) / block)), dim3(block), 0, ctx->stream, n, a,
                       x, b, y, out);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

// Wavenumber sums of the climate caller (climate.get_fluxes, reference picaso/climate.py:1931-1936): for every (level,
// profile) row  net = sum_w (plus - minus)[w] dwni[w].  One workgroup per row: 256 strided partial sums, then a fixed
// LDS tree -- deterministic, but not numpy's summation order (differences ~1e-16 relative to sum |terms|).
__global__ __launch_bounds__(256) void k_flux_net_sums(int nlevel, int nitem, int nwno, const double *__restrict__ plus,
                                                       const double *__restrict__ minus, const double *__restrict__ dw,
                                                       double *__restrict__ out)
{
#pragma clang fp contract(off)
    __shared__ double part[256];
    const int lev = blockIdx.x, item = blockIdx.y;
    const size_t row = ((size_t)lev * nitem + item) * nwno;          // rows are (level, profile * nwno + w)
    double acc = 0.0;
    for (int w = threadIdx.x; w < nwno; w += 256) acc = acc + (plus[row + w] - minus[row + w]) * dw[w];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] = part[threadIdx.x] + part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t)item * nlevel + lev] = part[0];
}

}  // namespace pz

extern "C" int picaso_flux_net_sums_dev(picaso_ctx *ctx, int nlevel, int nitem, int nwno, const double *disk4,
                                        const double *dwno, double *net_layer, double *net)
{
    using namespace pz;
    if (!ctx) return fail(nullptr, "null context");
    if (nlevel < 1 || nitem < 1 || nwno < 1 || !disk4 || !dwno || !net_layer || !net)
        return fail(ctx, "flux_net_sums: bad argument");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t plane = (size_t)nlevel * nitem * nwno;              // disk4: minus, plus, minus_mdpt, plus_mdpt
    const dim3 grid((unsigned)nlevel, (unsigned)nitem);
    hipLaunchKernelGGL(k_flux_net_sums, grid, dim3(256), 0, ctx->stream, nlevel, nitem, nwno, disk4 + 3 * plane,
                       disk4 + 2 * plane, dwno, net_layer);
    hipLaunchKernelGGL(k_flux_net_sums, grid, dim3(256), 0, ctx->stream, nlevel, nitem, nwno, disk4 + plane, disk4, dwno,
                       net);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}
