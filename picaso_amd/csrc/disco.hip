// Gauss-Chebyshev disk quadrature -- gfx950.
// Replaces disco.compress_disco / compress_thermal (reference picaso/disco.py:117-181).
// One lane per output element; the (g,t) sum runs in the reference's loop order so the result
// is deterministic and independent of how the wavelength grid is sharded.
#include "common.hpp"

namespace pz {


__global__ __launch_bounds__(256) void k_compress(size_t ninner, const double *__restrict__ x,
                                                  const double *__restrict__ wts, int nang,
                                                  const double *__restrict__ F0PI, double c1,
                                                  double c2, double *__restrict__ out)
{
    const size_t w = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (w >= ninner) return;
    double acc = 0.0;
    for (int k = 0; k < nang; ++k) acc = acc + x[(size_t)k * ninner + w] * wts[2 * k] * wts[2 * k + 1];
    // compress_disco: sym_fac*0.5*albedo/F0PI*(cos_theta+1)   (disco.py:148)
    // compress_thermal: flux*sym_fac                          (disco.py:181)
    out[w] = F0PI ? c1 * acc / F0PI[w] * c2 : acc * c1;
}

// wts_dev: device table of nang (gweight[g], tweight[t]) pairs in (g,t) loop order
int launch_compress_dev(picaso_ctx *ctx, size_t ninner, const double *x, const double *wts_dev,
                        int nang, const double *F0PI, double c1, double c2, double *out)
{
    if (ninner == 0) return 0;
    const int block = 256;
    const size_t grid = (ninner + block - 1) / block;
    hipLaunchKernelGGL(k_compress, dim3((unsigned)grid), dim3(block), 0, ctx->stream, ninner, x,
                       wts_dev, nang, F0PI, c1, c2, out);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace pz
