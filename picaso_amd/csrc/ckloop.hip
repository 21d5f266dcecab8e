// The correlated-k Gauss-point loop of the reference's picaso() around the spherical-harmonics solvers and around the
// 3-D solvers -- gfx950.
//
// Reference: picaso/justdoit.py:256-307 (reflected, `xint_at_top += xint*gauss_wts[ig]` over get_reflected_SH),
// :364-380 (thermal, get_thermal_SH), :488-516 (3-D: get_reflected_3d / get_thermal_3d once per Gauss point on
// `DTAU_3d[:, :, :, :, ig]`), :437-471 (compute_opacity facet by facet).  The reference slices plane[:, :, ig] out of the
// (nlayer|nlevel, nwno, ngauss) arrays of compute_opacity and calls the solver ngauss times.  Every solver on the path is
// pointwise in its column index, so here the Gauss axis IS the column axis: a plane (rows, nwno, ngauss) with the Gauss
// index fastest is a plane of nwno*ngauss columns for the existing kernels (one launch, 8 x the columns of a 661-bin
// k-table spectrum: a launch that fills the chip instead of eight that do not), the per-wavelength vectors (surface
// reflectivity, stellar flux, wavenumber) are repeated once per Gauss point, and k_weighted_colsum forms the reference's
// sum in ig order with one product and one sum per term (disco.hip).  No kernel of the solvers changes, hence a column's
// bits are those of the per-Gauss-point call.
//
// 3-D: the planes are FACET-MAJOR, (nfacets, nlayer|nlevel, nwno*ngauss), what picaso_compute_opacity_facet_major_ck_dev
// writes -- every facet is a spectrum of nwno*ngauss columns and one disk angle for the batched launches
// (picaso_get_reflected_3d_batch_dev / picaso_get_thermal_3d_batch_dev with numg = numt = 1): a wave holds 64 columns of
// one facet, all loads coalesced.  (The reference's (nlayer, nwno, ng, nt, ngauss) order would put a facet's Gauss points
// 8 doubles apart under the facet-fastest kernels: every load a 64 B stride.)
#include "common.hpp"

namespace pz {

__global__ __launch_bounds__(256) void k_repeat_cols(long ncol, int n, const double *__restrict__ in0,
                                                     const double *__restrict__ in1, const double *__restrict__ in2,
                                                     double *__restrict__ out0, double *__restrict__ out1,
                                                     double *__restrict__ out2)
{
    const long c = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    const long w = c / n;
    if (in0) out0[c] = in0[w];
    if (in1) out1[c] = in1[w];
    if (in2) out2[c] = in2[w];
}

// out_j[w*n + ig] = in_j[w]: up to three per-wavelength vectors at once (NULL inputs are skipped)
static int repeat_cols(picaso_ctx *ctx, long nwno, int n, const double *in0, double *out0, const double *in1 = nullptr,
                       double *out1 = nullptr, const double *in2 = nullptr, double *out2 = nullptr)
{
    const long ncol = nwno * n;
    hipLaunchKernelGGL(k_repeat_cols, dim3((unsigned)((ncol + 255) / 256)), dim3(256), 0, ctx->stream, ncol, n, in0, in1,
                       in2, out0, out1, out2);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

// a block of the context's pool for the lifetime of one call (released blocks are reused in stream order)
struct PoolBlock {
    picaso_ctx *ctx;
    void *p = nullptr;
    explicit PoolBlock(picaso_ctx *c) : ctx(c) {}
    int take(size_t bytes) { return picaso_dev_malloc(ctx, bytes, &p); }
    ~PoolBlock() { if (p) picaso_dev_free(ctx, p); }
    PoolBlock(const PoolBlock &) = delete;
    PoolBlock &operator=(const PoolBlock &) = delete;
};

static int check_ck(picaso_ctx *ctx, const char *who, int nlevel, int nwno, int ngauss, int numg, int numt,
                    const double *gauss_wts)
{
    if (ngauss < 1 || ngauss > MAX_CK_GAUSS) return fail(ctx, "%s: ngauss must be 1..%d, got %d", who, MAX_CK_GAUSS, ngauss);
    if (!gauss_wts) return fail(ctx, "%s: gauss_wts is null", who);
    if (nlevel < 2 || nwno < 1 || numg < 1 || numt < 1) return fail(ctx, "%s: bad sizes", who);
    // the inner solvers count columns in an int: nwno * ngauss (x facets in the 3-D forms) must fit
    if ((long)nwno * ngauss * (long)numg * numt > 2147483647L)
        return fail(ctx, "%s: nwno * ngauss * numg * numt = %ld columns exceed INT_MAX", who,
                    (long)nwno * ngauss * (long)numg * numt);
    return 0;
}

}  // namespace pz

using namespace pz;

extern "C" {

int picaso_get_reflected_SH_ck_dev(picaso_ctx *ctx, int nlevel, int nwno, int ngauss, int numg, int numt,
                                   const double *dtau, const double *tau, const double *w0, const double *cosb,
                                   const double *ftau_cld, const double *ftau_ray, const double *f_deltaM,
                                   const double *dtau_og, const double *tau_og, const double *w0_og,
                                   const double *cosb_og, const double *surf_reflect, const double *ubar0,
                                   const double *ubar1, double cos_theta, const double *F0PI, int w_single_form,
                                   int w_multi_form, int psingle_form, int w_single_rayleigh, int w_multi_rayleigh,
                                   int psingle_rayleigh, double frac_a, double frac_b, double frac_c,
                                   double constant_back, double constant_forward, int stream, double b_top,
                                   int single_form, int compound_f_deltaM, int cloud_free_above, const double *gauss_wts,
                                   double *xint_at_top, const double *gweight, const double *tweight, double *albedo)
{
    if (!ctx) return fail(nullptr, "null context");
    PZ_TRY(check_ck(ctx, "get_reflected_SH_ck", nlevel, nwno, ngauss, numg, numt, gauss_wts));
    if (!surf_reflect || !F0PI || !xint_at_top) return fail(ctx, "get_reflected_SH_ck: null argument");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const int nang = numg * numt;
    const long ncol = (long)nwno * ngauss;
    PoolBlock blk(ctx);
    PZ_TRY(blk.take(sizeof(double) * (size_t)ncol * (2 + (size_t)nang)));
    double *rs_c = (double *)blk.p, *f0_c = rs_c + ncol, *xcol = f0_c + ncol;
    PZ_TRY(repeat_cols(ctx, nwno, ngauss, surf_reflect, rs_c, F0PI, f0_c));
    // every Gauss point of every bin as a column of ONE solve (justdoit.py:256-269, all ig at once)
    PZ_TRY(picaso_get_reflected_SH_top_dev(ctx, nlevel, (int)ncol, ncol, numg, numt, dtau, tau, w0, cosb, ftau_cld, ftau_ray,
                                           f_deltaM, dtau_og, tau_og, w0_og, cosb_og, rs_c, ubar0, ubar1, cos_theta, f0_c,
                                           w_single_form, w_multi_form, psingle_form, w_single_rayleigh, w_multi_rayleigh,
                                           psingle_rayleigh, frac_a, frac_b, frac_c, constant_back, constant_forward,
                                           stream, b_top, 0, single_form, compound_f_deltaM, cloud_free_above, xcol,
                                           nullptr, nullptr, nullptr, nullptr));
    PZ_TRY(launch_weighted_colsum(ctx, nang, nwno, ngauss, gauss_wts, xcol, xint_at_top));       // justdoit.py:307
    if (albedo && gweight && tweight)
        PZ_TRY(picaso_compress_disco_dev(ctx, nwno, cos_theta, xint_at_top, gweight, numg, tweight, numt, F0PI, albedo));
    return 0;
}

int picaso_get_thermal_SH_ck_dev(picaso_ctx *ctx, int nlevel, const double *wno, int nwno, int ngauss, int numg,
                                 int numt, const double *tlevel, const double *dtau, const double *tau,
                                 const double *w0, const double *cosb_og, const double *plevel, const double *ubar1,
                                 const double *surf_reflect, int stream, int hard_surface,
                                 int cosb_differs_from_cosb_og, const double *gauss_wts, double *xint_at_top,
                                 const double *gweight, const double *tweight, double *flux_disk)
{
    if (!ctx) return fail(nullptr, "null context");
    PZ_TRY(check_ck(ctx, "get_thermal_SH_ck", nlevel, nwno, ngauss, numg, numt, gauss_wts));
    if (!wno || !surf_reflect || !xint_at_top) return fail(ctx, "get_thermal_SH_ck: null argument");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const int nang = numg * numt;
    const long ncol = (long)nwno * ngauss;
    PoolBlock blk(ctx);
    PZ_TRY(blk.take(sizeof(double) * (size_t)ncol * (2 + (size_t)nang)));
    double *rs_c = (double *)blk.p, *wn_c = rs_c + ncol, *xcol = wn_c + ncol;
    PZ_TRY(repeat_cols(ctx, nwno, ngauss, surf_reflect, rs_c, wno, wn_c));
    PZ_TRY(picaso_get_thermal_SH_dev(ctx, nlevel, wn_c, (int)ncol, ncol, numg, numt, tlevel, dtau, tau, w0, cosb_og, plevel,
                                     ubar1, rs_c, stream, hard_surface, cosb_differs_from_cosb_og, 0, xcol, nullptr,
                                     nullptr, nullptr));                                      // justdoit.py:364-370
    PZ_TRY(launch_weighted_colsum(ctx, nang, nwno, ngauss, gauss_wts, xcol, xint_at_top));       // justdoit.py:380
    if (flux_disk && gweight && tweight)
        PZ_TRY(picaso_compress_thermal_dev(ctx, (size_t)nwno, xint_at_top, gweight, numg, tweight, numt, flux_disk));
    return 0;
}

int picaso_get_reflected_3d_ck_dev(picaso_ctx *ctx, int nlevel, int nwno, int ngauss, int numg, int numt,
                                   const double *dtau, const double *tau, const double *w0, const double *cosb,
                                   const double *gcos2, const double *ftau_cld, const double *ftau_ray,
                                   const double *dtau_og, const double *tau_og, const double *w0_og,
                                   const double *cosb_og, const double *surf_reflect, const double *ubar0,
                                   const double *ubar1, double cos_theta, const double *F0PI, int single_phase,
                                   int multi_phase, double frac_a, double frac_b, double frac_c, double constant_back,
                                   double constant_forward, const double *gauss_wts, double *xint_at_top,
                                   const double *gweight, const double *tweight, double *albedo)
{
    if (!ctx) return fail(nullptr, "null context");
    PZ_TRY(check_ck(ctx, "get_reflected_3d_ck", nlevel, nwno, ngauss, numg, numt, gauss_wts));
    if (!dtau || !w0 || !surf_reflect || !F0PI || !ubar0 || !ubar1 || !xint_at_top)
        return fail(ctx, "get_reflected_3d_ck: null argument");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const int nfac = numg * numt, nlayer = nlevel - 1;
    const long ncol = (long)nwno * ngauss;
    PoolBlock blk(ctx);
    PZ_TRY(blk.take(sizeof(double) * (size_t)ncol * (2 + (size_t)nfac)));
    double *rs_c = (double *)blk.p, *f0_c = rs_c + ncol, *xcol = f0_c + ncol;
    PZ_TRY(repeat_cols(ctx, nwno, ngauss, surf_reflect, rs_c, F0PI, f0_c));
    // facet f = a spectrum of its own: one slab of every facet-major plane, one disk angle (its ubar0 / ubar1)
    const double *fam[11] = {dtau, tau, w0, cosb, gcos2, ftau_cld, ftau_ray, dtau_og, tau_og, w0_og, cosb_og};
    const bool level[11] = {false, true, false, false, false, false, false, false, true, false, false};
    std::vector<const double *> ptrs(13 * (size_t)nfac);
    const double *const *arg[11];
    for (int j = 0; j < 11; ++j) {
        if (!fam[j]) { arg[j] = nullptr; continue; }
        const size_t slab = (size_t)(level[j] ? nlevel : nlayer) * ncol;
        for (int f = 0; f < nfac; ++f) ptrs[(size_t)j * nfac + f] = fam[j] + slab * f;
        arg[j] = ptrs.data() + (size_t)j * nfac;
    }
    std::vector<double *> xs((size_t)nfac);
    std::vector<double> cts((size_t)nfac, cos_theta);
    for (int f = 0; f < nfac; ++f) {
        ptrs[(size_t)11 * nfac + f] = rs_c;
        ptrs[(size_t)12 * nfac + f] = f0_c;
        xs[f] = xcol + (size_t)f * ncol;
    }
    PZ_TRY(picaso_get_reflected_3d_batch_dev(ctx, nfac, nlevel, (int)ncol, 1, 1, arg[0], arg[1], arg[2], arg[3], arg[4], arg[5],
                                             arg[6], arg[7], arg[8], arg[9], arg[10], ptrs.data() + (size_t)11 * nfac, ubar0,
                                             ubar1, cts.data(), ptrs.data() + (size_t)12 * nfac, single_phase, multi_phase,
                                             frac_a, frac_b, frac_c, constant_back, constant_forward, xs.data(), nullptr,
                                             nullptr, nullptr));                              // justdoit.py:488-497
    PZ_TRY(launch_weighted_colsum(ctx, nfac, nwno, ngauss, gauss_wts, xcol, xint_at_top));       // justdoit.py:498
    if (albedo && gweight && tweight)
        PZ_TRY(picaso_compress_disco_dev(ctx, nwno, cos_theta, xint_at_top, gweight, numg, tweight, numt, F0PI, albedo));
    return 0;
}

int picaso_get_thermal_3d_ck_dev(picaso_ctx *ctx, int nlevel, const double *wno, int nwno, int ngauss, int numg,
                                 int numt, const double *tlevel_3d, const double *dtau, const double *w0,
                                 const double *cosb, const double *plevel_3d, const double *ubar1,
                                 const double *surf_reflect, int hard_surface, const double *gauss_wts,
                                 double *int_at_top, const double *gweight, const double *tweight, double *flux_disk)
{
    if (!ctx) return fail(nullptr, "null context");
    PZ_TRY(check_ck(ctx, "get_thermal_3d_ck", nlevel, nwno, ngauss, numg, numt, gauss_wts));
    if (!wno || !dtau || !w0 || !surf_reflect || !tlevel_3d || !plevel_3d || !ubar1 || !int_at_top)
        return fail(ctx, "get_thermal_3d_ck: null argument");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const int nfac = numg * numt, nlayer = nlevel - 1;
    const long ncol = (long)nwno * ngauss;
    PoolBlock blk(ctx);
    PZ_TRY(blk.take(sizeof(double) * (size_t)ncol * (2 + (size_t)nfac)));
    double *rs_c = (double *)blk.p, *wn_c = rs_c + ncol, *xcol = wn_c + ncol;
    PZ_TRY(repeat_cols(ctx, nwno, ngauss, surf_reflect, rs_c, wno, wn_c));
    const size_t slab = (size_t)nlayer * ncol;
    std::vector<const double *> ptrs(4 * (size_t)nfac);
    std::vector<double *> xs((size_t)nfac);
    // level tables (nlevel, numg, numt) -> one (nlevel) column per pseudo-spectrum
    std::vector<double> tl((size_t)nfac * nlevel), pl((size_t)nfac * nlevel);
    for (int f = 0; f < nfac; ++f) {
        ptrs[f] = dtau + slab * f;
        ptrs[(size_t)nfac + f] = w0 + slab * f;
        ptrs[(size_t)2 * nfac + f] = cosb ? cosb + slab * f : nullptr;
        ptrs[(size_t)3 * nfac + f] = rs_c;
        xs[f] = xcol + (size_t)f * ncol;
        for (int l = 0; l < nlevel; ++l) {
            tl[(size_t)f * nlevel + l] = tlevel_3d[(size_t)l * nfac + f];
            pl[(size_t)f * nlevel + l] = plevel_3d[(size_t)l * nfac + f];
        }
    }
    PZ_TRY(picaso_get_thermal_3d_batch_dev(ctx, nfac, nlevel, wn_c, (int)ncol, 1, 1, tl.data(), ptrs.data(),
                                           ptrs.data() + nfac, cosb ? ptrs.data() + (size_t)2 * nfac : nullptr, pl.data(),
                                           ubar1, ptrs.data() + (size_t)3 * nfac, hard_surface, xs.data(), nullptr, nullptr,
                                           nullptr));                                         // justdoit.py:502-513
    PZ_TRY(launch_weighted_colsum(ctx, nfac, nwno, ngauss, gauss_wts, xcol, int_at_top));        // justdoit.py:514
    if (flux_disk && gweight && tweight)
        PZ_TRY(picaso_compress_thermal_dev(ctx, (size_t)nwno, int_at_top, gweight, numg, tweight, numt, flux_disk));
    return 0;
}

}  // extern "C"
