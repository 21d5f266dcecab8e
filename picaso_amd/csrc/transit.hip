// Transmission spectrum (Rp/Rs)^2 -- gfx950.
// Replaces fluxes.get_transit_1d (reference picaso/fluxes.py:2581-2663; Brown 2001 eq. 11).
//
// Per wavelength: slant optical depth along the chord tangent at every level i,
//     TAUALL_i = sum_{j<i} 2 TAU[i-j-1] delta_length[i][j],   TAU[l] = DTAU[l]/colden[l]*mmw[l],
// then F = (min z / Rs)^2 + 2/Rs^2 sum_i (1 - exp(-TAUALL_i)) z_i dz_i.  The chord geometry
// delta_length (nlevel x nlevel, wavelength independent; fluxes.py:2625-2644) is formed on the host
// in the reference's arithmetic and read as wave-uniform scalars; the kernel is the O(nlevel^2)
// per-wavelength part.  One lane per wavelength, 64 lanes per block: the block's slice of the scaled
// optical-depth plane is staged once in LDS ([layer][lane], conflict free), so HBM sees each DTAU
// element exactly once (algorithmic bytes 8 nwno (nlayer + 1)).
#include "common.hpp"
#include "device_math.hpp"

namespace pz {

struct TransitArgs {
    int nlevel, nwno;
    long pitch;
    const double *dtau;        // (nlayer, nwno) device
    const double *tab;         // device table: delta_length[nlevel*nlevel], zdz[nlevel], colden[nlayer], mmw_g[nlayer]
    double zmin_term, two_over_rs2;
    double *out;               // (nwno)
};

constexpr int TRANSIT_BLOCK = 64;
#ifndef PZ_TRANSIT_ROWS
#define PZ_TRANSIT_ROWS 2
#endif
constexpr int TRANSIT_ROWS = PZ_TRANSIT_ROWS;
#ifndef PZ_TRANSIT_WAVES
#define PZ_TRANSIT_WAVES 4
#endif
constexpr int TRANSIT_WAVES = PZ_TRANSIT_WAVES;

// TRANSIT_WAVES waves share one tile of 64 wavelengths: wave v sums the chords i = v, v + WAVES, ... (chord i has i terms:
// round-robin keeps the waves level), leaves (1 - exp(-TAUALL_i)) z_i dz_i in the tile's place once every wave is done
// with it, and wave 0 adds those terms in level order -- the reference's order, whoever formed them.  With one wave per
// tile a CU held three waves (the tile is 46 KB) and the kernel waited on the latency of its dependent fp64 adds.
__global__ __launch_bounds__(TRANSIT_BLOCK * TRANSIT_WAVES) void k_transit(const TransitArgs a)
{
    extern __shared__ double tau_lds[];                 // [nlevel][64]: nlayer rows of TAU, later nlevel rows of terms
    const int lane = threadIdx.x % TRANSIT_BLOCK, wave = threadIdx.x / TRANSIT_BLOCK;
    const long w = blockIdx.x * (long)TRANSIT_BLOCK + lane;
    const int n = a.nlevel, nl = n - 1;
    const double *dl = a.tab, *zdz = dl + (long)n * n, *colden = zdz + n, *mmw = colden + nl;
    const bool live = w < a.nwno;
    for (int l = wave; l < nl; l += TRANSIT_WAVES) {    // TAU = DTAU / colden * mmw   (fluxes.py:2648-2650)
        const double d = live ? a.dtau[(long)l * a.pitch + w] : 0.0;
        tau_lds[l * TRANSIT_BLOCK + lane] = d / colden[l] * mmw[l];
    }
    __syncthreads();
    // A chord's sum is one dependent chain of fp64 adds in the reference's order (j ascending); TRANSIT_ROWS chords of a
    // wave are summed side by side -- independent chains, each still in its own order, so the same bits.
    // spectrum('transmission') at 1e5 x 90, one wave per tile: 0.83 ms with one chord at a time, 0.67 with two, 0.73 with
    // four, 0.75 with eight, 0.91 with sixteen (the chord geometry arrives through scalar loads, one per chord and shell).
    // With TRANSIT_WAVES waves per tile (two chords side by side): 2 waves 0.57 ms, 4 waves 0.48, 8 waves 0.47.
    const double *const tl = tau_lds + lane;
    constexpr int STEP = TRANSIT_WAVES;
    constexpr int MAXT = 64;                            // chords per wave the registers hold: nlevel <= MAXT * WAVES
    double term[MAXT];
    int nt = 0;
    for (int i0 = wave; i0 < n; i0 += STEP * TRANSIT_ROWS) {
        double t[TRANSIT_ROWS];
#pragma unroll
        for (int k = 0; k < TRANSIT_ROWS; ++k) t[k] = 0.0;
        if (i0 + STEP * (TRANSIT_ROWS - 1) < n) {
            for (int j = 0; j < i0; ++j) {              // every chord of the group crosses shell j
#pragma unroll
                for (int k = 0; k < TRANSIT_ROWS; ++k)  // two because of the sphere's symmetry (:2655-2656)
                    t[k] = t[k] + (2.0 * tl[(i0 + STEP * k - j - 1) * TRANSIT_BLOCK]) * dl[(long)(i0 + STEP * k) * n + j];
            }
#pragma unroll
            for (int k = 1; k < TRANSIT_ROWS; ++k)      // the shells only the deeper chords of the group reach
                for (int j = i0; j < i0 + STEP * k; ++j)
                    t[k] = t[k] + (2.0 * tl[(i0 + STEP * k - j - 1) * TRANSIT_BLOCK]) * dl[(long)(i0 + STEP * k) * n + j];
        } else {                                        // the last, short group: one chord at a time
            for (int k = 0; i0 + STEP * k < n; ++k)
                for (int j = 0; j < i0 + STEP * k; ++j)
                    t[k] = t[k] + (2.0 * tl[(i0 + STEP * k - j - 1) * TRANSIT_BLOCK]) * dl[(long)(i0 + STEP * k) * n + j];
        }
#pragma unroll
        for (int k = 0; k < TRANSIT_ROWS; ++k)
            if (i0 + STEP * k < n && nt < MAXT) term[nt++] = (1.0 - fexp(-t[k])) * zdz[i0 + STEP * k];   // (:2660-2661)
    }
    __syncthreads();                                    // every wave is done with TAU: the tile now takes the terms
    {
        int q = 0;
        for (int i = wave; i < n; i += STEP) tau_lds[i * TRANSIT_BLOCK + lane] = term[q++];
    }
    __syncthreads();
    if (wave != 0 || !live) return;
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc = acc + tl[i * TRANSIT_BLOCK];           // the disk sum in level order
    a.out[w] = a.zmin_term + a.two_over_rs2 * acc;
}

int launch_transit(picaso_ctx *ctx, const TransitArgs &a)
{
    const size_t lds = sizeof(double) * (size_t)a.nlevel * TRANSIT_BLOCK;
    if (lds > 160 * 1024 || a.nlevel > 64 * TRANSIT_WAVES)
        return fail(ctx, "get_transit_1d: %d levels exceed the LDS tile", a.nlevel);
    const long grid = (a.nwno + TRANSIT_BLOCK - 1) / TRANSIT_BLOCK;
    if (lds > 64 * 1024)
        PZ_HIP(ctx, hipFuncSetAttribute((const void *)k_transit, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_transit, dim3((unsigned)grid), dim3(TRANSIT_BLOCK * TRANSIT_WAVES), lds, ctx->stream, a);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace pz

using namespace pz;

extern "C" {

int picaso_get_transit_1d_dev(picaso_ctx *ctx, const double *z, const double *dz, int nlevel, int nwno,
                              long plane_pitch, double rstar, const double *mmw, double k_b, double amu,
                              const double *player, const double *tlayer, const double *colden,
                              const double *dtau, double *rprs2)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nlevel < 2 || nwno < 1) return fail(ctx, "get_transit_1d: bad sizes nlevel=%d nwno=%d", nlevel, nwno);
    PZ_NEED(ctx, "get_transit_1d", z, dz, mmw, player, tlayer, colden, dtau, rprs2);
    if (plane_pitch < nwno) return fail(ctx, "get_transit_1d: plane_pitch %ld < nwno %d", plane_pitch, nwno);
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const int n = nlevel, nl = nlevel - 1;
    std::vector<double> tab((size_t)n * n + n + 2 * (size_t)nl, 0.0);
    double *dlen = tab.data(), *zdz = dlen + (size_t)n * n, *cd = zdz + n, *mg = cd + nl;
    // chord segments between shells (fluxes.py:2625-2644); `player`/`tlayer` are indexed exactly as
    // the reference indexes them (its caller passes the LEVEL pressure/temperature, justdoit.py:391-393)
    double seg = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) {
            const double ref = z[i], inner = z[i - j], outer = z[i - j - 1];
            if (inner != ref && outer != ref) seg = sqrt(outer * outer - ref * ref) - sqrt(inner * inner - ref * ref);
            else if (inner == ref) seg = sqrt(outer * outer - ref * ref);
            dlen[(size_t)i * n + j] = seg * player[i - j - 1] / tlayer[i - j - 1] / k_b;
        }
    double zmin = z[0];
    for (int i = 0; i < n; ++i) { zdz[i] = z[i] * dz[i]; zmin = z[i] < zmin ? z[i] : zmin; }
    for (int l = 0; l < nl; ++l) { cd[l] = colden[l]; mg[l] = mmw[l] * amu; }   // mmw in grams (:2623)
    const void *d_tab = nullptr;
    PZ_TRY(table_upload(ctx, tab.data(), sizeof(double) * tab.size(), &d_tab));
    TransitArgs a{};
    a.nlevel = nlevel; a.nwno = nwno; a.pitch = plane_pitch; a.dtau = dtau; a.tab = (const double *)d_tab;
    a.zmin_term = (zmin / rstar) * (zmin / rstar);
    a.two_over_rs2 = 2.0 / (rstar * rstar);
    a.out = rprs2;
    return launch_transit(ctx, a);
}

int picaso_get_transit_1d(picaso_ctx *ctx, const double *z, const double *dz, int nlevel, int nwno,
                          double rstar, const double *mmw, double k_b, double amu, const double *player,
                          const double *tlayer, const double *colden, const double *dtau, double *rprs2)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nlevel < 2 || nwno < 1) return fail(ctx, "get_transit_1d: bad sizes nlevel=%d nwno=%d", nlevel, nwno);
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nl = (size_t)(nlevel - 1) * nwno;
    PZ_TRY(arena_reset(ctx, sizeof(double) * (nl + (size_t)nwno) + 8 * 256));
    const double *d_dtau;
    PZ_TRY(arena_upload(ctx, dtau, nl, &d_dtau));
    double *d_out = (double *)arena_take(ctx, sizeof(double) * (size_t)nwno);
    if (!d_out) return fail(ctx, "arena exhausted");
    PZ_TRY(picaso_get_transit_1d_dev(ctx, z, dz, nlevel, nwno, nwno, rstar, mmw, k_b, amu, player, tlayer, colden,
                                     d_dtau, d_out));
    PZ_HIP(ctx, hipMemcpyAsync(rprs2, d_out, sizeof(double) * (size_t)nwno, hipMemcpyDeviceToHost, ctx->stream));
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int picaso_get_transit_1d_ck_dev(picaso_ctx *ctx, const double *z, const double *dz, int nlevel, int nwno,
                                 int ngauss, double rstar, const double *mmw, double k_b, double amu,
                                 const double *player, const double *tlayer, const double *colden,
                                 const double *dtau, const double *gauss_wts, double *rprs2)
{
    if (!ctx) return fail(nullptr, "null context");
    if (ngauss < 1 || ngauss > MAX_CK_GAUSS) return fail(ctx, "get_transit_1d_ck: ngauss must be 1..%d", MAX_CK_GAUSS);
    if (!gauss_wts) return fail(ctx, "get_transit_1d_ck: gauss_wts is null");
    if (nlevel < 2 || nwno < 1) return fail(ctx, "get_transit_1d_ck: bad sizes nlevel=%d nwno=%d", nlevel, nwno);
    const long ncol = (long)nwno * ngauss;
    if (ncol > 0x7fffffffL) return fail(ctx, "get_transit_1d_ck: nwno*ngauss exceeds 2^31-1");
    PZ_TRY(ck_scratch_reserve(ctx, sizeof(double) * (size_t)ncol));
    PZ_TRY(picaso_get_transit_1d_dev(ctx, z, dz, nlevel, (int)ncol, ncol, rstar, mmw, k_b, amu, player, tlayer,
                                     colden, dtau, ctx->ck_scratch));
    // rprs2 += rprs2_g * gauss_wts[ig] in ig order (justdoit.py:388-405)
    return launch_weighted_colsum(ctx, 1, nwno, ngauss, gauss_wts, ctx->ck_scratch, rprs2);
}

}  // extern "C"
